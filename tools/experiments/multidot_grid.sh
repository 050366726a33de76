#!/bin/bash
# multi_dot: elements of x per thread (grid = n / (256 * pt), at most 2048 workgroups)
R=${GRAFT_REPO_ROOT:-$PWD}
N=${1:-10000000}; shift
for pt in ${@:-2 8 16 32 64}; do echo "== N $N elements per thread $pt"; HIPEIG_MULTIDOT_PER_THREAD=$pt python3 $R/tools/blas_bench.py /tmp/x.json $N 2>&1 >/dev/null | grep "multi_dot\|cgs2"; done
