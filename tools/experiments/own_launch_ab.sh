#!/bin/bash
# One rank's sweeps alone (bench.py phases.sweeps_alone_ms) in a 6-process rehearsal: own windows in a launch of their
# own or riding in the first chunk's launch, with one and two exchange chunks.
R=${GRAFT_REPO_ROOT:-$PWD}
export HIPEIG_COMM=direct
for P in 6 4; do for own in 1 0; do for ch in 2 1; do
  echo "== ranks $P own-window launch $own exchange chunks $ch"
  HIPEIG_OWN_LAUNCH=$own HIPEIG_GATHER_CHUNKS=$ch timeout -k 10 200 python3 $R/bench.py --gpus $P --steps 20 --warmup 5 --no-lanczos 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["phases"])'
done; done; done
