#!/bin/bash
# N = 1e6 on the aligned layout: rows per row block (one or two workgroups per CU)
R=${GRAFT_REPO_ROOT:-$PWD}
for rw in 1344 1984 2688 3968; do echo "== rows per block $rw"; HIPEIG_TCOOW_RW=$rw timeout -k 10 100 python3 $R/bench.py --n 1000000 --nnz-row 32 --steps 300 --warmup 20 --no-cpu --no-lanczos --no-block 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/product", d["config"]["layout"]["row_blocks"], d["config"]["layout"]["workgroups_per_launch"])'; done
