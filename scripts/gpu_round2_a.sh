#!/bin/bash
# first GPU pass of round 2: parity tests, then bench A/B of the kernel variants
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r2_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r2_tests.log
tail -5 gpurun_out/r2_tests.log
for v in "8 2" "8 1" "4 4" "8 4" "16 1" "16 2"; do
  set -- $v
  timeout -k 10 300 python bench.py --lanes $1 --gpw $2 --steps 60 --warmup 20 --no-cpu > gpurun_out/r2_bench_$1_$2.json 2> gpurun_out/r2_bench_$1_$2.err
  echo "variant $1x$2 rc=$?"; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r2_bench_$1_$2.json").read().strip().splitlines()[-1])
    print("  value %.1f M env-steps/s, %.3f ms/step, kernel %.3f ms" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
except Exception as e:
    print("  no result", e)
PY
done
