#!/bin/bash
# Same-box A/B of two builds of libmonsoon_hip (box-to-box variance is ~4 %): alternates bench.py runs.
#   gpurun -- bash scripts/ab_bench.sh monsoon_amd/libmonsoon_hip_A.so monsoon_amd/libmonsoon_hip.so [rounds]
A=$1; B=$2; R=${3:-2}
for i in $(seq $R); do
  for L in $A $B; do
    MONSOON_LIB=$(pwd)/$L timeout -k 10 200 python bench.py --no-cpu > gpurun_out/ab_tmp.log 2>&1 || { echo "FAILED $L"; tail -5 gpurun_out/ab_tmp.log; exit 1; }
    python - "$L" <<'PY'
import json,sys
r=json.loads(open("gpurun_out/ab_tmp.log").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:44s} {r['value']/1e6:8.1f} M env-steps/s  {r['ms_per_step']:.4f} ms/step", flush=True)
PY
  done
done
