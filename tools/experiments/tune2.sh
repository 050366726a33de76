#!/bin/bash
N=${1:-10000000}; R=${2:-64}
run() { out=$(env "$@" timeout -k 10 120 python bench.py --n $N --nnz-row $R --steps 10 --warmup 2 --no-cpu --no-lanczos --variant 3 2>&1 | tail -1)
  echo "$* :: $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["roofline"]["avg_launch_ms"], "ms/launch frac", round(d["value"]/8000,4))' 2>/dev/null || echo "$out" | cut -c1-300)"; }
run HIPEIG_TCOO_PREFETCH=0
run HIPEIG_TCOO_PREFETCH=1
run HIPEIG_TCOO_PREFETCH=1 HIPEIG_TCOO_WBITS=17
run HIPEIG_TCOO_PREFETCH=0 HIPEIG_TCOO_WBITS=17
run HIPEIG_TCOO_PREFETCH=1 HIPEIG_TCOO_WBITS=19
run HIPEIG_TCOO_PREFETCH=1 HIPEIG_TCOO_RW=1280 HIPEIG_TCOO_WG_PER_CU=4
run HIPEIG_TCOO_PREFETCH=1 HIPEIG_TCOO_RW=1280 HIPEIG_TCOO_WG_PER_CU=4 HIPEIG_TCOO_WBITS=17
