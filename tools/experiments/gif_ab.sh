#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
for lib in libhipeig.so libhipeig_gif4.so libhipeig_gif2.so; do for cs in 1 5; do
  echo "== $lib csplit $cs N=1e6"; HIPEIG_LIB=$R/eigensolvers_amd/$lib HIPEIG_TCOOW_CSPLIT=$cs timeout -k 10 100 python3 $R/bench.py --n 1000000 --nnz-row 32 --steps 200 --warmup 20 --no-cpu --no-lanczos --no-block 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/product")'
done
echo "== $lib N=1e7"; HIPEIG_LIB=$R/eigensolvers_amd/$lib timeout -k 10 100 python3 $R/bench.py --steps 20 --warmup 5 --no-cpu --no-lanczos --no-block 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/product")'
done
