#!/bin/bash
# One run each (no retries): (1) stand-alone capture/replay under rocprofv3 --kernel-trace, four capture shapes;
# (2) the library's own captured MINRES chunk (HIPEIG_GRAPH=1) under rocprofv3 --kernel-trace.
# Everything the tool prints goes to gpurun_out/graph_rocprof_*.txt.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for args in "1 158 1" "0 158 1" "1 8 1" "1 158 0"; do
  tag=$(echo $args | tr ' ' '_')
  echo "== plain run: graph_repro $args" > $O/graph_rocprof_repro_$tag.txt
  $R/tools/graph_repro $args >> $O/graph_rocprof_repro_$tag.txt 2>&1; echo "rc=$?" >> $O/graph_rocprof_repro_$tag.txt
  echo "== rocprofv3 --kernel-trace: graph_repro $args" >> $O/graph_rocprof_repro_$tag.txt
  rocprofv3 --kernel-trace --output-format csv -d $O/graph_prof_$tag -- $R/tools/graph_repro $args >> $O/graph_rocprof_repro_$tag.txt 2>&1
  echo "rc=$?" >> $O/graph_rocprof_repro_$tag.txt
  tail -3 $O/graph_rocprof_repro_$tag.txt
done
echo "== library, HIPEIG_GRAPH=1, plain" > $O/graph_rocprof_lib.txt
HIPEIG_GRAPH=1 python3 $R/tools/experiments/graph_solve.py >> $O/graph_rocprof_lib.txt 2>&1; echo "rc=$?" >> $O/graph_rocprof_lib.txt
echo "== library, HIPEIG_GRAPH=1, rocprofv3 --kernel-trace" >> $O/graph_rocprof_lib.txt
export HIPEIG_GRAPH=1
rocprofv3 --kernel-trace --output-format csv -d $O/graph_prof_lib -- python3 $R/tools/experiments/graph_solve.py >> $O/graph_rocprof_lib.txt 2>&1
echo "rc=$?" >> $O/graph_rocprof_lib.txt
grep -E "rc=|ok|iterations|Error|error|abort|Abort|signal|core" $O/graph_rocprof_lib.txt | head -20
