#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
for pt in 4 8 16 32 64; do echo "== per_thread $pt"; HIPEIG_MRB_PER_THREAD=$pt python3 $R/tools/block_bench.py --variants 2 2>/dev/null | python3 -c 'import sys,json; d=json.load(sys.stdin); print(d["block_solve"]["ms_per_block_iteration"], d["block_vs_single_speedup"], d["block_solve"]["iterations"][:3])'; done
