#!/bin/bash
# A/B of library builds on the block benchmark: block_ab.sh lib1.so lib2.so ...   (HIPEIG_LIB picks the build)
for round in 1 2; do
 for lib in "$@"; do
  echo "== $lib (round $round)"
  HIPEIG_LIB=$PWD/eigensolvers_amd/$lib timeout -k 10 120 python tools/block_bench.py --variants 2 2>&1 | grep -E "ms_incl|ms_per_block_iteration|speedup"
 done
done
