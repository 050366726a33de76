#!/bin/bash
# A/B two builds of the library on the same box, interleaved rounds.
N=${1:-10000000}; R=${2:-64}
for round in 1 2 3; do
 for lib in eigensolvers_amd/libhipeig_old.so eigensolvers_amd/libhipeig.so; do
  out=$(HIPEIG_LIB=$PWD/$lib timeout -k 10 120 python bench.py --n $N --nnz-row $R --steps 10 --warmup 2 --no-cpu --no-lanczos 2>&1 | tail -1)
  echo "$lib :: $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step frac", d["roofline"]["frac"])' 2>/dev/null || echo "$out" | cut -c1-300)"
 done
done
