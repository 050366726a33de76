#!/bin/bash
# window size / row-block sweep of the window-blocked block product (results: gpurun_out/block_sweep.txt)
for wb in ${WBLIST:-8 9 10 11 12}; do
  echo "== HIPEIG_BCOO_WBITS=$wb"
  HIPEIG_BCOO_WBITS=$wb timeout -k 10 120 python tools/block_bench.py --no-solve --variants 2 2>&1 | grep -E "ms_incl|windows|rows_per_block"
done
