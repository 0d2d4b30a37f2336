#!/bin/bash
# same-box A/B: round-1 library vs the variants of the current build
mkdir -p gpurun_out
run() {  # name, env..., -- args
  local name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-cpu $EXTRA > gpurun_out/r2b_$name.json 2> gpurun_out/r2b_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r2b_$name.json").read().strip().splitlines()[-1])
    print("$name: %.1f M env-steps/s, kernel %.3f ms" % (d["value"]/1e6, d["roofline"]["avg_launch_ms"]))
except Exception as e:
    print("$name: no result", e)
PY
}
run r1 MONSOON_LIB=monsoon_amd/libmonsoon_hip_r1.so
run v814 MONSOON_LANES=8 MONSOON_GPW=1 MONSOON_WPE=4
run v815 MONSOON_LANES=8 MONSOON_GPW=1 MONSOON_WPE=5
run v816 MONSOON_LANES=8 MONSOON_GPW=1 MONSOON_WPE=6
run v416 MONSOON_LANES=4 MONSOON_GPW=1 MONSOON_WPE=6
run v418 MONSOON_LANES=4 MONSOON_GPW=1 MONSOON_WPE=8
run r1_again MONSOON_LIB=monsoon_amd/libmonsoon_hip_r1.so
run v814_again MONSOON_LANES=8 MONSOON_GPW=1 MONSOON_WPE=4
