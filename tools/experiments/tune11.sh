#!/bin/bash
N=${1:-10000000}; R=${2:-64}
run() { out=$(env "$@" timeout -k 10 120 python bench.py --size $N --nnz-row $R --steps 10 --warmup 2 --no-cpu --no-lanczos 2>&1 | tail -1)
  echo "$* :: $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step", d["roofline"]["launches_per_step"], "launches")' 2>/dev/null || echo "$out" | cut -c1-300)"; }
run A=0
run HIPEIG_TCOOW_RW=10112
run HIPEIG_TCOOW_RW=13440
run HIPEIG_TCOOW_RW=6720
