#!/bin/bash
# N = 1e6 on the aligned layout: full-height row blocks shared by `csplit` workgroups (k = 10.7 instead of 2.1 touches per
# line) + the combine launch, against one short row block per CU
R=${GRAFT_REPO_ROOT:-$PWD}
for cs in 1 3 5 8; do echo "== csplit $cs"; HIPEIG_TCOOW_CSPLIT=$cs python3 $R/tools/experiments/minres_iter_time.py 1000000 32 | grep "fuse_kd 1" | tail -1
  HIPEIG_TCOOW_CSPLIT=$cs timeout -k 10 100 python3 $R/bench.py --n 1000000 --nnz-row 32 --steps 200 --warmup 20 --no-cpu --no-lanczos --no-block 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/product", d["config"]["layout"])'
done
