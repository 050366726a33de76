#!/bin/bash
# Folded column-split launches: sweep tasks drawn from one queue per XCD (own XCD first) against one queue for all.
R=${GRAFT_REPO_ROOT:-$PWD}
prod() { timeout -k 10 100 python3 $R/bench.py --n 1000000 --nnz-row 32 --steps 200 --warmup 20 --no-cpu --no-lanczos --no-block 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); l=d["config"]["layout"]; print(d["ms_per_step"], "ms/product", {k: l.get(k) for k in ("rows_per_block","row_blocks","column_splits","gathers_in_flight","epilogue_tasks_per_block")})'; }
mr() { python3 $R/tools/experiments/minres_iter_time.py 1000000 32 | grep "fuse_kd 1" | tail -1; }
for rep in 1 2; do for nq in 8 1; do for cs in 5 4 8; do
  echo "== N = 1e6, queues $nq, column splits $cs"; HIPEIG_TCOOW_FOLD_QUEUES=$nq HIPEIG_TCOOW_CSPLIT=$cs prod; HIPEIG_TCOOW_FOLD_QUEUES=$nq HIPEIG_TCOOW_CSPLIT=$cs mr
done; done; done
echo "== N = 1e6, no splits"; HIPEIG_TCOOW_CSPLIT=1 prod; HIPEIG_TCOOW_CSPLIT=1 mr
for P in 4 8; do for nq in 8 1; do
  echo "== slab of a P = $P run, queues $nq"; HIPEIG_TCOOW_FOLD_QUEUES=$nq python3 $R/tools/experiments/slab_time.py $P
done; echo "== slab of a P = $P run, round 3"; HIPEIG_LIB=$R/eigensolvers_amd/libhipeig_r3.so python3 $R/tools/experiments/slab_time.py $P; done
