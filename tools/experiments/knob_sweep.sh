#!/bin/bash
# One parameterised sweep instead of a script per experiment: every argument is one setting of the
# library's tuning knobs (quoted, space-separated VAR=value pairs); each is run through bench.py on the
# headline configuration and reported as "settings :: ms/step frac".
#   tools/experiments/knob_sweep.sh "HIPEIG_TCOOW_BINBITS=5" "HIPEIG_TCOOW_BINBITS=4 HIPEIG_TCOOW_WBITS=18" ...
# (round 1 ran it over HIPEIG_TCOOW_BINBITS 3..6, HIPEIG_TCOOW_WBITS 15..18, HIPEIG_TCOOW_RW, HIPEIG_TCOOW_CSPLIT,
#  HIPEIG_TCOOW_UNCACHED and HIPEIG_OVERLAP; results are quoted in DESIGN.md section 3.1.)
N=${N:-10000000}; R=${R:-64}; EXTRA=${EXTRA:-}
for setting in "$@"; do
  out=$(env $setting timeout -k 10 200 python bench.py --n $N --nnz-row $R --steps 10 --warmup 2 --no-cpu --no-lanczos $EXTRA 2>/dev/null | tail -1)
  echo "$setting :: $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step frac", d["roofline"]["frac"])' 2>/dev/null || echo "$out" | cut -c1-200)"
done
