#!/bin/bash
# Vector-L1 (TCP) counters of the block product kernels at N = 1e6, k = 8 (window-blocked and row-owner): is the 64-byte
# operand gather of 4 lanes x 16 B one tag access or four?  usage (inside gpurun): bash tools/experiments/pmc_tcp_block.sh
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; TAG=${1:-tcpb}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc_${TAG}_$i -- python3 $R/tools/block_bench.py --n 1000000 --nnz-row 32 --k 8 --variants 1,2 --no-solve --reps 5 > $O/pmc_${TAG}_$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - <<PY
import csv, glob, collections
for i in (1, 2, 3):
    acc = collections.defaultdict(list)
    for f in glob.glob("$O/pmc_${TAG}_%d/**/*_counter_collection.csv" % i, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "spmm_" in k or "spmv_tcoow" in k:
                acc[(k.split("(")[0][-28:], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(k[0], k[1], "mean per launch %.4g over %d launches" % (sum(v) / len(v), len(v)))
PY
