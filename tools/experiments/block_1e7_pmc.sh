#!/bin/bash
# HBM traffic of the k = 8 block product at N = 1e7, 64 nnz/row (row-owner kernel): does a 64-byte operand gather
# cost a whole 128-byte line?  Two PMC passes + a kernel trace.  usage (inside gpurun): bash tools/experiments/block_1e7_pmc.sh
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="$R/tools/block_bench.py --n 10000000 --nnz-row 64 --k 8 --variants 1 --no-solve --reps 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b1e7_trace -- python3 $ARGS > $O/b1e7_trace.log 2>&1
tail -1 $O/b1e7_trace.log | cut -c1-600
for pass in "fetch:FETCH_SIZE" "req:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "l2:TCC_HIT_sum TCC_MISS_sum"; do
  name=${pass%%:*}; ctr=${pass#*:}
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $O/b1e7_$name -- python3 $ARGS > $O/b1e7_$name.log 2>&1
  echo "pmc $name rc=$?"
done
python3 - <<PY
import csv, glob, collections
for name in ("fetch", "req", "l2"):
    for f in glob.glob("$O/b1e7_%s/**/*counter_collection.csv" % name, recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:60], r["Counter_Name"])
            agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
        for k, (n, v) in sorted(agg.items()):
            if "rowowner" in k[0] or "spmm" in k[0]:
                print(name, k, "launches", n, "per launch %.4g" % (v / n))
PY
