#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r2c_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r2c_tests.log
run() {
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python bench.py --no-cpu "$@" > gpurun_out/r2c_$name.json 2> gpurun_out/r2c_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r2c_$name.json").read().strip().splitlines()[-1])
    print("$name: %.1f M env-steps/s, kernel %.3f ms, %.1f look/dec" % (d["value"]/1e6, d["roofline"]["avg_launch_ms"], d["lookahead_per_decision"]))
except Exception as e:
    print("$name: no result", e)
PY
}
run r1 MONSOON_LIB=monsoon_amd/libmonsoon_hip_r1.so -- --rounds 1 --steps 96 --warmup 24
run w5_r1 X=1 -- --rounds 1 --steps 96 --warmup 24
run w5_r8 X=1 -- --rounds 8 --steps 12 --warmup 3
run w4_r8 MONSOON_WPE=4 -- --rounds 8 --steps 12 --warmup 3
run w6_r8 MONSOON_WPE=6 -- --rounds 8 --steps 12 --warmup 3
run u4w8_r8 MONSOON_LANES=4 MONSOON_WPE=8 -- --rounds 8 --steps 12 --warmup 3
run u16w3_r8 MONSOON_LANES=16 MONSOON_WPE=3 -- --rounds 8 --steps 12 --warmup 3
run w5_r8_again X=1 -- --rounds 8 --steps 12 --warmup 3
