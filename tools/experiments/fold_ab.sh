#!/bin/bash
# Round 4: column-split products whose slabs are added by epilogue workgroups of the SAME launch (no combine launch),
# against the round-3 library (libhipeig_r3.so: raw launch + combine launch).  N = 1e6 (configs #2/#3) and the slabs a rank
# of a 4- / 8-GPU run of N = 1e7 owns.
R=${GRAFT_REPO_ROOT:-$PWD}
prod() { timeout -k 10 100 python3 $R/bench.py --n 1000000 --nnz-row 32 --steps 200 --warmup 20 --no-cpu --no-lanczos --no-block 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); l=d["config"]["layout"]; print(d["ms_per_step"], "ms/product", {k: l.get(k) for k in ("rows_per_block","row_blocks","column_splits","gathers_in_flight","epilogue_tasks_per_block")})'; }
mr() { python3 $R/tools/experiments/minres_iter_time.py 1000000 32 | grep "fuse_kd 1" | tail -1; }
echo "== round-3 library, shipped choice (no splits at N = 1e6)"; HIPEIG_LIB=$R/eigensolvers_amd/libhipeig_r3.so prod; HIPEIG_LIB=$R/eigensolvers_amd/libhipeig_r3.so mr
echo "== round-3 library, 5 column splits + combine launch"; HIPEIG_TCOOW_CSPLIT=5 HIPEIG_LIB=$R/eigensolvers_amd/libhipeig_r3.so prod; HIPEIG_TCOOW_CSPLIT=5 HIPEIG_LIB=$R/eigensolvers_amd/libhipeig_r3.so mr
echo "== new library, default (splits + 4 gathers in flight, folded)"; prod; mr
echo "== new library, no splits"; HIPEIG_TCOOW_CSPLIT=1 prod; HIPEIG_TCOOW_CSPLIT=1 mr
echo "== new library, splits, one gather in flight"; HIPEIG_TCOOW_GIF=1 prod; HIPEIG_TCOOW_GIF=1 mr
for epu in 1 2 10 20; do echo "== new library, $epu epilogue tasks per row block"; HIPEIG_TCOOW_EPU=$epu prod; HIPEIG_TCOOW_EPU=$epu mr; done
for cs in 3 4 8 10; do echo "== new library, $cs column splits"; HIPEIG_TCOOW_CSPLIT=$cs prod; HIPEIG_TCOOW_CSPLIT=$cs mr; done
for P in 2 4 8; do
  echo "== slab of a P = $P run: round 3"; HIPEIG_LIB=$R/eigensolvers_amd/libhipeig_r3.so python3 $R/tools/experiments/slab_time.py $P
  echo "== slab of a P = $P run: folded"; python3 $R/tools/experiments/slab_time.py $P
done
