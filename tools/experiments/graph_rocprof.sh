#!/bin/bash
# rocprofv3 --kernel-trace against captured hipGraphs (profiles/r02_hipgraph_under_rocprofv3.txt).  One run per configuration,
# no retries.  usage: graph_rocprof.sh 1|2|3   (1: capture shapes + the library; 2: narrowing down inside bench.py; 3: replay count)
STAGE=${1:-1}
if [ "$STAGE" = 1 ]; then
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
elif [ "$STAGE" = 2 ]; then
# Narrowing down the rocprofv3 + captured-MINRES-chunk crash: distinct configurations, ONE run each.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export HIPEIG_GRAPH=1
run() {  # tag, rocprof flags, env..., -- args
  tag=$1; flags=$2; shift 2
  echo "== $tag: rocprofv3 $flags, env: $ENVS, args: $*" > $O/graph2_$tag.txt
  env $ENVS timeout -k 10 200 rocprofv3 $flags --output-format csv -d $O/graph2_prof_$tag -- python3 $R/tools/experiments/graph_solve.py "$@" >> $O/graph2_$tag.txt 2>&1
  echo "rc=$?" >> $O/graph2_$tag.txt
  echo "$tag: $(grep -E '^rc=|^graph ' $O/graph2_$tag.txt | tr '\n' ' ')"
}
export HIPEIG_GRAPH_TRACE=1
ENVS="PRE_GRAM=1" run e9_1e7_pregram "--kernel-trace" 10000000 64
echo "== e10: bench.py --no-cpu --no-block, HIPEIG_GRAPH=1 HIPEIG_GRAPH_TRACE=1, rocprofv3 --kernel-trace" > $O/graph2_e10_bench_trace.txt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/graph2_prof_e10 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-block >> $O/graph2_e10_bench_trace.txt 2>&1
echo "rc=$?" >> $O/graph2_e10_bench_trace.txt; grep -E "hipeig graph|^rc=" $O/graph2_e10_bench_trace.txt | tail -8
else
# Hypothesis test (one run each): the abort needs a few hundred replays of a captured graph (the AQL queue
# ring wrapping under the profiler's packet interception), nothing of libhipeig.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
echo "== plain: graph_repro 1 158 1 24 1500" > $O/graph3_repro.txt
$R/tools/graph_repro 1 158 1 24 1500 >> $O/graph3_repro.txt 2>&1; echo "rc=$?" >> $O/graph3_repro.txt
echo "== rocprofv3 --kernel-trace: graph_repro 1 158 1 24 1500" >> $O/graph3_repro.txt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/graph3_prof -- $R/tools/graph_repro 1 158 1 24 1500 >> $O/graph3_repro.txt 2>&1; echo "rc=$?" >> $O/graph3_repro.txt
grep -v simple_timer $O/graph3_repro.txt | grep -E "^==|^rc=|ok, record|replay [0-9]*00 done|SIGSEGV|Aborted" | tail -30
fi
