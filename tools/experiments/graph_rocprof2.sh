#!/bin/bash
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
