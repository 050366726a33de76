#!/bin/bash
# Round 3, the one SpMV experiment left (VERDICT r2, item 5 ii): bins aligned to 64-element instruction groups
# (HIPEIG_TCOOW_ALIGN=1) against the shipped layout: time per product and L1 -> L2 read requests per launch.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out
for al in 0 1 0 1; do
  echo "== HIPEIG_TCOOW_ALIGN=$al"
  HIPEIG_TCOOW_ALIGN=$al timeout -k 10 200 python3 $R/bench.py --steps 30 --warmup 5 --no-cpu --no-lanczos --no-block 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step frac", d["roofline"]["frac"], "median", d["roofline"]["single_step_ms_median"])'
done
cd /tmp && export TMPDIR=/tmp
for al in 0 1; do
  HIPEIG_TCOOW_ALIGN=$al timeout -k 10 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $O/pmc_align_$al -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu --no-lanczos --no-block > $O/pmc_align_$al.log 2>&1
  echo "pmc align=$al rc=$?"
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/pmc_align_$al/**/*_counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "spmv_tcoow_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print("align=$al", k, "mean per launch %.5g over %d launches" % (sum(v) / len(v), len(v)))
PY
done
