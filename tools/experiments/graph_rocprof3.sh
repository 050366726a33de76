#!/bin/bash
# Hypothesis test (one run each): the abort needs a few hundred replays of a captured graph (the AQL queue
# ring wrapping under the profiler's packet interception), nothing of libhipeig.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
echo "== plain: graph_repro 1 158 1 24 1500" > $O/graph3_repro.txt
$R/tools/graph_repro 1 158 1 24 1500 >> $O/graph3_repro.txt 2>&1; echo "rc=$?" >> $O/graph3_repro.txt
echo "== rocprofv3 --kernel-trace: graph_repro 1 158 1 24 1500" >> $O/graph3_repro.txt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/graph3_prof -- $R/tools/graph_repro 1 158 1 24 1500 >> $O/graph3_repro.txt 2>&1; echo "rc=$?" >> $O/graph3_repro.txt
grep -v simple_timer $O/graph3_repro.txt | grep -E "^==|^rc=|ok, record|replay [0-9]*00 done|SIGSEGV|Aborted" | tail -30
