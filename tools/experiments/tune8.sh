#!/bin/bash
N=${1:-10000000}; R=${2:-64}
run() { out=$(env "$@" timeout -k 10 120 python bench.py --n $N --nnz-row $R --steps 10 --warmup 2 --no-cpu --no-lanczos 2>&1 | tail -1)
  echo "$* :: $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step frac", d["roofline"]["frac"])' 2>/dev/null || echo "$out" | cut -c1-300)"; }
for r in 1 2; do
run A=0
run HIPEIG_TCOOW_UNCACHED=1
run HIPEIG_TCOOW_UNCACHED=1 HIPEIG_TCOO_ABLATE=1
run HIPEIG_TCOO_ABLATE=1
done
