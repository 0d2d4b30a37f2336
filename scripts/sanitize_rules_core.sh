#!/bin/bash
# AddressSanitizer + UBSan run of the PRODUCT's rules core (the explicit work stack of monsoon_amd/csrc/rules.h: the same
# sources the device compiles, built for the host with the device's 21-word resident stack so that the eviction path
# runs too) -- GPU sanitizers are not available on this pool.  MSB_SAN_CORE=oracle runs the recursive oracle instead.
set -e
cd "$(dirname "$0")/../oracle"
CORE=${MSB_SAN_CORE:-product}
D=""; [ "$CORE" = product ] && D="-DORC_PRODUCT_CORE -DMSB_HOST_SKW=21"
F="-O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -fsanitize=address,undefined -fno-sanitize-recover=undefined -Wno-psabi -I../monsoon_amd/csrc $D -shared"
g++ $F -o /tmp/liboracle_asan.so oracle.cpp -lpthread
g++ $F -DMSB_EXT=1 -o /tmp/liboracle_ext_asan.so oracle.cpp -lpthread
g++ $F -DMSB_EXT=2 -o /tmp/liboracle_big_asan.so oracle.cpp -lpthread
cd ..
echo "sanitizing the $CORE core"
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
  UBSAN_OPTIONS=print_stacktrace=1 python scripts/sanitize_rules_core.py
