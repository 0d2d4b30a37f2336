#!/bin/bash
# Experiment builds for scripts/ab_libs.sh: the default hot-kernel variant (k_play_reg, 8 lanes, 5 waves/SIMD) recompiled with other
# flags and linked with the product build's other objects.   bash scripts/ab_build.sh name "flags" [name "flags" ...]
# A flag beginning with '~' removes that word from the product flags ("~-enable-ipra=0" drops `-mllvm -enable-ipra=0`).
set -e
cd "$(dirname "$0")/../monsoon_amd/csrc"
make -s -j8 ../libmonsoon_hip.so
BASE="-O3 -std=c++17 -ffp-contract=off -fno-strict-aliasing -mllvm -disable-promote-alloca-to-lds=true -mllvm -enable-ipra=0 -fPIC -Wno-unused-value"
OTHERS=$(ls build/std/*.o | grep -v variant_8_5_10.o)
while [ $# -ge 2 ]; do
  name=$1; extra=$2; shift 2
  flags=$BASE; add=""
  for w in $extra; do
    case $w in
      "~"*) flags=${flags/-mllvm ${w#\~}/} ;;
      *) add="$add $w" ;;
    esac
  done
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 $flags $add -DVAR_U=8 -DVAR_W=5 -DVAR_G=10 -c -o build/exp_$name.o variant.hip 2> build/exp_$name.err \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmonsoon_hip_$name.so $OTHERS build/exp_$name.o && echo "built $name" ) &
  while [ $(jobs -r | wc -l) -ge 8 ]; do sleep 1; done
done
wait
