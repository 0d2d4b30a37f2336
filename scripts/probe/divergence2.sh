#!/bin/bash
# gpurun -- bash scripts/probe/divergence2.sh   (kernel trace of scripts/probe/divergence2.py per API_LANES build)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/divergence2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for L in 64 32 16; do
  lib=$ROOT/monsoon_amd/libmonsoon_hip_l$L.so; [ $L = 64 ] && lib=$ROOT/monsoon_amd/libmonsoon_hip.so
  export MONSOON_LIB=$lib
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/t$L -- python3 $ROOT/scripts/probe/divergence2.py > $OUT/run$L.txt 2> $OUT/run$L.err || { tail -5 $OUT/run$L.err; exit 1; }
  cat $OUT/run$L.txt
  python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/t$L/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_step" in r["Kernel_Name"]]
    for r in rows[-4:]:
        print("  lanes $L k_step %.3f ms  LDS %s scratch %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("LDS_Block_Size"), r.get("Scratch_Size")))
PY
done
