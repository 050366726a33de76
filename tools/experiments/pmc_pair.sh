#!/bin/bash
# L1 counters of the pair sweep forced on at N = 1e7 next to the single sweep (why the 16-byte form loses there).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export HIPEIG_PAIR_SWEEP=1
timeout -k 10 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum --output-format csv -d $O/pmc_pair -- python3 $R/tools/pair_bench.py ${1:-10000000} > $O/pmc_pair.log 2>&1
echo "rc=$?"
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc_pair/**/*_counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "tcoow" in k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    for c, v in cs.items():
        print(k, c, "mean per launch %.4g over %d launches" % (sum(v) / len(v), len(v)))
PY
