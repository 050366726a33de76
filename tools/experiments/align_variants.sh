#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
for round in 1 2; do for lib in libhipeig.so libhipeig_u2.so libhipeig_u8.so libhipeig_il.so libhipeig_t512.so; do
  echo "== $lib"; HIPEIG_LIB=$R/eigensolvers_amd/$lib timeout -k 10 200 python3 $R/bench.py --steps 30 --warmup 5 --no-cpu --no-lanczos --no-block 2>&1 | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step frac", d["roofline"]["frac"], "median", d["roofline"]["single_step_ms_median"])'
done; done
