#!/bin/bash
N=${1:-10000000}; R=${2:-64}
run() { v=$1; shift; out=$(env "$@" timeout -k 10 120 python bench.py --n $N --nnz-row $R --steps 10 --warmup 2 --no-cpu --no-lanczos --variant $v 2>&1 | tail -1)
  echo "variant=$v $* :: $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["roofline"]["avg_launch_ms"], "ms/step frac", round(d["value"]/8000,4))' 2>/dev/null || echo "$out" | cut -c1-300)"; }
run 4 A=0
run 4 HIPEIG_TCOO_ABLATE=1
run 4 HIPEIG_TCOO_ABLATE=2
run 4 HIPEIG_TCOOW_BINBITS=6
run 4 HIPEIG_TCOOW_WBITS=16
