#!/bin/bash
mkdir -p gpurun_out
run() {
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python bench.py --no-cpu "$@" > gpurun_out/r2d_$name.json 2> gpurun_out/r2d_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r2d_$name.json").read().strip().splitlines()[-1])
    print("$name: %.1f M env-steps/s, kernel %.3f ms, cap faults %d" % (d["value"]/1e6, d["roofline"]["avg_launch_ms"], d["capacity_faults"]))
except Exception as e:
    print("$name: no result", e)
PY
}
for st in 16384 4096 5120 6000 8192 10000 12288 17000 24576; do
  run stack_$st X=1 -- --rounds 8 --steps 12 --warmup 3 --stack $st
done
run w4_stack6000 MONSOON_WPE=4 -- --rounds 8 --steps 12 --warmup 3 --stack 6000
