#!/bin/bash
N=${1:-10000000}; R=${2:-64}
run() { out=$(env HIPEIG_LIB=$PWD/eigensolvers_amd/libhipeig_exp.so "$@" timeout -k 10 120 python bench.py --size $N --nnz-row $R --steps 10 --warmup 2 --no-cpu --no-lanczos 2>&1 | tail -1)
  echo "$* :: $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step")' 2>/dev/null || echo "$out" | cut -c1-300)"; }
run A=0
run HIPEIG_TCOO_ABLATE=7
run HIPEIG_TCOO_ABLATE=3
run HIPEIG_TCOO_ABLATE=6
run HIPEIG_TCOO_ABLATE=2
run HIPEIG_TCOO_ABLATE=5
