#!/bin/bash
# A/B of the row-owner block kernel's unroll (gathers in flight per wave) at N = 1e7, 64 nnz/row, k = 8 and at N = 1e6
for lib in libhipeig_ro1.so libhipeig_ro2.so libhipeig.so libhipeig_ro8.so; do
  echo "== $lib"
  HIPEIG_LIB=$PWD/eigensolvers_amd/$lib timeout -k 10 200 python tools/block_bench.py --n 10000000 --nnz-row 64 --k 8 --variants 1 --no-solve --reps 5 2>&1 | tr ',' '\n' | grep -E "ms_incl|spmv_ms"
  HIPEIG_LIB=$PWD/eigensolvers_amd/$lib timeout -k 10 200 python tools/block_bench.py --n 1000000 --nnz-row 32 --k 8 --variants 1 --no-solve --reps 10 2>&1 | tr ',' '\n' | grep -E "ms_incl"
done
