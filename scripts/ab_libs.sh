#!/bin/bash
# Same-box A/B of alternative builds of the library: bash scripts/ab_libs.sh <lib suffix> ...  (monsoon_amd/libmonsoon_hip_<suffix>.so;
# "std" = the product library).  Experiment builds are linked by hand from monsoon_amd/csrc/build/ objects.
mkdir -p gpurun_out
for x in "$@"; do
  lib=monsoon_amd/libmonsoon_hip_$x.so; [ "$x" = std ] && lib=monsoon_amd/libmonsoon_hip.so
  MONSOON_LIB=$lib timeout -k 10 300 python bench.py --no-cpu > gpurun_out/ab_$x.json 2> gpurun_out/ab_$x.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab_$x.json").read().strip().splitlines()[-1])
    print("$x: %.1f M env-steps/s, kernel %.3f ms" % (d["value"]/1e6, d["roofline"]["avg_launch_ms"]))
except Exception as e:
    print("$x: no result", e)
PY
done
