#!/bin/bash
# fused MGS sweep: elements per thread (grid = n / (256 * pt) workgroups, records of 2 values summed on two levels)
# usage: mgs_grid.sh N pt...
R=${GRAFT_REPO_ROOT:-$PWD}
N=${1:-10000000}; shift
for pt in ${@:-2 4 8 16 32 48 64 96 128}; do echo "== N $N elements per thread $pt"; HIPEIG_MGS_PER_THREAD=$pt python3 $R/tools/blas_bench.py /tmp/x.json $N 2>&1 >/dev/null | grep " mgs"; done
