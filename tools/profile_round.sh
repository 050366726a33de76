#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats, then three separate PMC passes (HBM reads, HBM
# writes, L2 hit/miss) as MI355X_MICROARCH.md prescribes.  Raw output under gpurun_out/; condense it
# afterwards with tools/summarize_profiles.py <tag> gpurun_out/prof_<tag> gpurun_out/pmc_<tag>_fetch ...
# usage (inside gpurun): bash tools/profile_round.sh <tag>
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -- python3 $R/bench.py > $O/prof_$TAG.log 2>&1      # the default command, as the driver runs it
tail -1 $O/prof_$TAG.log | cut -c1-300
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "l2:TCC_HIT_sum TCC_MISS_sum"; do
  name=${pass%%:*}; ctr=${pass#*:}
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_${TAG}_$name -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-lanczos --no-block > $O/pmc_${TAG}_$name.log 2>&1
  echo "pmc $name rc=$?"
done
