#!/bin/bash
# MINRES on a column-split sweep: rows per thread of the combine launch (the launch that adds the slabs and carries the
# KA epilogue with the riding KD).  N = 1e6 with 5 splits forced.
R=${GRAFT_REPO_ROOT:-$PWD}
for ct in 24 12 8 4 2; do echo "== combine rows per thread $ct"; HIPEIG_TCOOW_CSPLIT=5 HIPEIG_MR_COMBINE_PER_THREAD=$ct python3 $R/tools/experiments/minres_iter_time.py 1000000 32 | grep "fuse_kd 1" | tail -1; done
echo "== no splits"; python3 $R/tools/experiments/minres_iter_time.py 1000000 32 | grep "fuse_kd 1" | tail -1
