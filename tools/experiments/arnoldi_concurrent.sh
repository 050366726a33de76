#!/bin/bash
# Would the blocked Arnoldi sweeps of SEVERAL lock-step solves gain from running concurrently (own streams)?  Emulated with P
# processes (each its own context) that start their timed loops at the same wall-clock second and run for seconds:
# per-process time per step, i.e. aggregate = that / P.  usage: arnoldi_concurrent.sh N m reps
R=${GRAFT_REPO_ROOT:-$PWD}; N=${1:-1000000}; M=${2:-28}; REPS=${3:-3000}
for P in 1 2 4; do
  echo "== $P concurrent processes (N $N, $REPS steps each)"
  export ARN_START_AT=$(( $(date +%s) + 45 ))
  for i in $(seq 1 $P); do python3 $R/tools/experiments/arnoldi_bench.py $N $M $REPS 2>&1 | grep "cols 4" | tail -1 & done
  wait
done
