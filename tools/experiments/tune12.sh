#!/bin/bash
# bin width / window width re-check of the TCOO-W layout with the final kernel
run() { out=$(env "$@" timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu --no-lanczos 2>/dev/null | tail -1); echo "$* :: $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms frac", d["roofline"]["frac"])' 2>/dev/null)"; }
run HIPEIG_TCOOW_BINBITS=5
run HIPEIG_TCOOW_BINBITS=4
run HIPEIG_TCOOW_BINBITS=3
run HIPEIG_TCOOW_BINBITS=4 HIPEIG_TCOOW_WBITS=18
run HIPEIG_TCOOW_BINBITS=5
run HIPEIG_TCOOW_BINBITS=4
