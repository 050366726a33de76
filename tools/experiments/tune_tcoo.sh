#!/bin/bash
# Sweep the TCOO tuning knobs on the headline configuration (run on the GPU box).
N=${1:-10000000}; R=${2:-64}
for wb in 16 17 18 19; do
  for cfg in "2560 2" "1280 4" "1280 2" "640 8" "640 4"; do
    set -- $cfg
    out=$(HIPEIG_TCOO_WBITS=$wb HIPEIG_TCOO_RW=$1 HIPEIG_TCOO_WG_PER_CU=$2 timeout -k 10 120 python bench.py --n $N --nnz-row $R --steps 10 --warmup 2 --no-cpu --no-lanczos --variant 3 2>&1 | tail -1)
    echo "wbits=$wb rw=$1 wg_per_cu=$2 :: $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["roofline"]["avg_launch_ms"], "ms/launch", d["ms_per_step"], "ms/step frac", round(d["value"]/8000,4))' 2>/dev/null || echo "$out" | cut -c1-200)"
  done
done
