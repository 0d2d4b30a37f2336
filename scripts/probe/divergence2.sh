#!/bin/bash
# gpurun -- bash scripts/probe/divergence2.sh   (kernel trace of scripts/probe/divergence2.py per API_LANES build)
# The 32- and 16-lane builds of the API kernels are probe builds (monsoon_hip.hip with -DMSB_API_LANES), made here if missing.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for L in 32 16; do
  if [ ! -f $ROOT/monsoon_amd/libmonsoon_hip_l$L.so ]; then
    ( cd $ROOT/monsoon_amd/csrc && make -s -j8 ../libmonsoon_hip.so && mkdir -p build/x &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-strict-aliasing -mllvm -disable-promote-alloca-to-lds=true \
        -mllvm -enable-ipra=0 -fPIC -Wno-unused-value -DMSB_API_LANES=$L -c -o build/x/main_l$L.o monsoon_hip.hip &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmonsoon_hip_l$L.so build/x/main_l$L.o build/std/variant_*.o ) || exit 1
  fi
done
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
