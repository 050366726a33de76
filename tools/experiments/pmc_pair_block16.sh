#!/bin/bash
# Vector-L1 / L2 counters of the block complex-shift product with 4 and with 8 complex operands per pass (8- / 16-wide
# blocks), window-blocked kernel forced.  usage (inside gpurun): bash tools/experiments/pmc_pair_block16.sh N nnz_row
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; N=${1:-1000000}; NR=${2:-32}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/pb16.py <<PY
import os, sys
sys.path.insert(0, "$R")
import numpy as np, eigensolvers_amd as ea
N, R, W = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
os.environ["HIPEIG_PAIR_BLOCK_WIDTH"] = W
ctx = ea.HipContext.default()
H = ea.HipCsrOperator.generate(N, R, seed=7)
H.set_block_variant(2)
xs = [(ctx.alloc(N), ctx.alloc(N)) for _ in range(8)]
for a, b in xs:
    ea._lib.call("hipeig_vec_fill", ctx.handle, a.ptr, N, 1.0); ea._lib.call("hipeig_vec_fill", ctx.handle, b.ptr, N, 0.5)
for _ in range(3):
    H.apply_shifted_pairs(0.05 + 0.11j, xs)
ctx.synchronize()
PY
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  for W in 4 8; do
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc_pb16_${N}_${W}_$i -- python3 /tmp/pb16.py $N $NR $W > $O/pmc_pb16_${N}_${W}_$i.log 2>&1
    echo "pass $i width $W rc=$?"
  done
done
python3 - <<PY
import csv, glob, collections
for W in (4, 8):
    acc = collections.defaultdict(list)
    for i in (1, 2):
        for f in glob.glob("$O/pmc_pb16_${N}_%d_%d/**/*_counter_collection.csv" % (W, i), recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                if "spmm_bcoo" in k:
                    acc[(k.split("(")[0][5:60], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("N $N, %d complex operands per pass:" % W, k[0], k[1], "mean per launch %.4g over %d launches" % (sum(v) / len(v), len(v)))
PY
