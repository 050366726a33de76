#!/bin/bash
# TCP (vector L1) counters of the SpMV sweep: read-request latency, stall causes, translation misses.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; TAG=${1:-tcp}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc_${TAG}_$i -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu --no-lanczos > $O/pmc_${TAG}_$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - <<PY
import csv, glob, collections
for i in (1, 2, 3):
    acc = collections.defaultdict(list)
    for f in glob.glob("$O/pmc_${TAG}_%d/**/*_counter_collection.csv" % i, recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Kernel_Name"].startswith("spmv_tcoow_kernel"):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print(k, "mean per launch %.4g over %d launches" % (sum(v) / len(v), len(v)))
PY
