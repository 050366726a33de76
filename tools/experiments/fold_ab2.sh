#!/bin/bash
# N = 1e6: the round-3 library with 5 column splits + combine launch + four gathers in flight (build flag) against the folded
# launch of this round, and both against no splits; then the slabs.
R=${GRAFT_REPO_ROOT:-$PWD}
prod() { timeout -k 10 100 python3 $R/bench.py --n 1000000 --nnz-row 32 --steps 200 --warmup 20 --no-cpu --no-lanczos --no-block 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); l=d["config"]["layout"]; print(d["ms_per_step"], "ms/product", {k: l.get(k) for k in ("rows_per_block","row_blocks","column_splits","gathers_in_flight","epilogue_tasks_per_block")})'; }
mr() { python3 $R/tools/experiments/minres_iter_time.py 1000000 32 | grep "fuse_kd 1" | tail -1; }
echo "== round 3, no splits"; HIPEIG_LIB=$R/eigensolvers_amd/libhipeig_r3.so prod; HIPEIG_LIB=$R/eigensolvers_amd/libhipeig_r3.so mr
echo "== round 3 + 4 gathers in flight, 5 splits + combine launch"; HIPEIG_TCOOW_CSPLIT=5 HIPEIG_LIB=$R/eigensolvers_amd/libhipeig_r3gif4.so prod; HIPEIG_TCOOW_CSPLIT=5 HIPEIG_LIB=$R/eigensolvers_amd/libhipeig_r3gif4.so mr
echo "== round 3 + 4 gathers in flight, 4 splits + combine launch"; HIPEIG_TCOOW_CSPLIT=4 HIPEIG_LIB=$R/eigensolvers_amd/libhipeig_r3gif4.so prod; HIPEIG_TCOOW_CSPLIT=4 HIPEIG_LIB=$R/eigensolvers_amd/libhipeig_r3gif4.so mr
echo "== folded, default"; prod; mr
echo "== folded, 4 splits"; HIPEIG_TCOOW_CSPLIT=4 prod; HIPEIG_TCOOW_CSPLIT=4 mr
echo "== folded, 5 splits, 2 epilogue tasks per row block"; HIPEIG_TCOOW_EPU=2 prod; HIPEIG_TCOOW_EPU=2 mr
echo "== folded, 5 splits, 3 epilogue tasks per row block"; HIPEIG_TCOOW_EPU=3 prod; HIPEIG_TCOOW_EPU=3 mr
echo "== no splits"; HIPEIG_TCOOW_CSPLIT=1 prod; HIPEIG_TCOOW_CSPLIT=1 mr
for P in 4 8; do
  echo "== slab of a P = $P run, folded"; python3 $R/tools/experiments/slab_time.py $P
  echo "== slab of a P = $P run, round 3"; HIPEIG_LIB=$R/eigensolvers_amd/libhipeig_r3.so python3 $R/tools/experiments/slab_time.py $P; done
