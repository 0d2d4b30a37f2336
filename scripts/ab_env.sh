#!/bin/bash
# Same-box comparison of environment-variable variants of one build:  bash scripts/ab_env.sh "VAR=1" "VAR=2 OTHER=3" ...
for i in 1 2; do
  for V in "$@"; do
    env $V timeout -k 10 200 python bench.py --no-cpu > gpurun_out/ab_tmp.log 2>&1 || { echo "FAILED $V"; tail -5 gpurun_out/ab_tmp.log; exit 1; }
    python - "$V" <<'PY'
import json,sys
r=json.loads(open("gpurun_out/ab_tmp.log").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:44s} {r['value']/1e6:8.1f} M env-steps/s  {r['ms_per_step']:.4f} ms/step", flush=True)
PY
  done
done
