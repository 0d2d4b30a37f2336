#!/bin/bash
# One validation pass on the GPU box (gpurun -- bash scripts/gpu_validate.sh [tag]): the default bench line (C2, with the CPU
# baseline), the C3 / C4 / C5 generation lines, whole-rollout timings, the C3 GA loop.  Everything lands under gpurun_out/validate_<tag>_*.
TAG=${1:-r03}
mkdir -p gpurun_out
O=gpurun_out/validate_${TAG}
timeout -k 10 300 python bench.py > ${O}_bench.json 2> ${O}_bench.err; echo "bench rc=$?"
for w in c3 c4 c5; do
  timeout -k 10 300 python bench.py --workload $w > ${O}_bench_$w.json 2> ${O}_bench_$w.err; echo "bench $w rc=$?"
done
for d in "16384 N12M" "65536 N12M" "16384 S12" "65536 S12" "16384 random"; do
  set -- $d
  timeout -k 10 300 python scripts/rollout_timing.py $1 $2 > ${O}_rollout_$1_$2.log 2>&1; tail -2 ${O}_rollout_$1_$2.log
done
timeout -k 10 300 python scripts/run_c3.py > ${O}_run_c3.json 2> ${O}_run_c3.err; echo "run_c3 rc=$?"
python - <<PY
import json
d = json.loads(open("${O}_bench.json").read().strip().splitlines()[-1])
print("bench: %.1f M env-steps/s, %.3f ms/step, k_play %.3f ms, roofline frac %.3f, cpu baseline %.1f M on %d cores" % (
    d["value"] / 1e6, d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["cpu_baseline"]["value"] / 1e6, d["cpu_baseline"]["cores"]))
for w in ("c3", "c4", "c5"):
    d = json.loads(open("${O}_bench_%s.json" % w).read().strip().splitlines()[-1])
    ho = d["roofline"]["host_and_other_ms_per_step"]
    print("bench %s: %.1f M env-steps/s end to end, %.1f ms per generation (k_play %.1f ms summed over the tiers' handles, host + other %s), tiers %s" % (
        w, d["value"] / 1e6, d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], "%.1f ms" % ho if ho is not None else "n/a: the tiers overlap", d["record_tiers"]))
PY
