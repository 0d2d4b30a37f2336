#!/bin/bash
# AddressSanitizer + UBSan run of the rules core (CPU build of the same sources the device compiles).
set -e
cd "$(dirname "$0")/../oracle"
F="-O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -fsanitize=address,undefined -fno-sanitize-recover=undefined -Wno-psabi -shared"
g++ $F -o /tmp/liboracle_asan.so oracle.cpp -lpthread
g++ $F -DMSB_EXT=1 -o /tmp/liboracle_ext_asan.so oracle.cpp -lpthread
g++ $F -DMSB_EXT=2 -o /tmp/liboracle_big_asan.so oracle.cpp -lpthread
cd ..
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
  UBSAN_OPTIONS=print_stacktrace=1 python scripts/sanitize_rules_core.py
