#!/bin/bash
# blocked Arnoldi sweep at N = 1e6 (where FEAST config #5 spends 67 % of its time): threads per workgroup x workgroups
R=${GRAFT_REPO_ROOT:-$PWD}
for lib in libhipeig.so libhipeig_at512.so libhipeig_at1024.so; do for g in 128 256 512 768; do
  echo "== $lib wgs $g"; HIPEIG_LIB=$R/eigensolvers_amd/$lib HIPEIG_ARNOLDI_WGS=$g python3 $R/tools/experiments/arnoldi_bench.py 1000000 28 12 | tail -1
done; done
