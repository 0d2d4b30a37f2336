#!/bin/bash
# One validation pass on the GPU box (gpurun -- bash scripts/gpu_validate.sh): parity tests, the C5 slice against the
# CPU replay, rollout timings, the default bench line.  Everything lands under gpurun_out/validate_*.
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/validate_tests.log 2>&1
rc=$?
echo "gpu tests rc=$rc"; tail -3 gpurun_out/validate_tests.log
[ $rc -ne 0 ] && exit $rc
python scripts/c5_parity.py 32768 > gpurun_out/validate_c5.log 2>&1; echo "c5 rc=$?"; tail -1 gpurun_out/validate_c5.log
for d in "16384 N12M" "65536 N12M" "16384 S12" "16384 random"; do
  set -- $d
  python scripts/rollout_timing.py $1 $2 > gpurun_out/validate_rollout_$1_$2.log 2>&1; tail -2 gpurun_out/validate_rollout_$1_$2.log
done
python bench.py > gpurun_out/validate_bench.json 2> gpurun_out/validate_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/validate_bench.json").read().strip().splitlines()[-1])
print("bench: %.1f M env-steps/s, %.3f ms/step, k_play %.3f ms, roofline frac %.3f, cpu baseline %.1f M on %d cores" % (
    d["value"] / 1e6, d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["cpu_baseline"]["value"] / 1e6, d["cpu_baseline"]["cores"]))
PY
