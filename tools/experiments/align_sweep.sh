#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
run() { env "$@" timeout -k 10 200 python3 $R/bench.py --steps 30 --warmup 5 --no-cpu --no-lanczos --no-block 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step frac", d["roofline"]["frac"], "median", d["roofline"]["single_step_ms_median"])'; }
for bb in 4 5 6 7; do echo "== align=1 binbits=$bb"; run HIPEIG_TCOOW_ALIGN=1 HIPEIG_TCOOW_BINBITS=$bb; done
for wb in 15 16; do echo "== align=1 wbits=$wb"; run HIPEIG_TCOOW_ALIGN=1 HIPEIG_TCOOW_WBITS=$wb; done
echo "== align=1 N=1e6"; env HIPEIG_TCOOW_ALIGN=1 timeout -k 10 200 python3 $R/tools/experiments/minres_iter_time.py 1000000 32 | tail -2
echo "== align=0 N=1e6"; env HIPEIG_TCOOW_ALIGN=0 timeout -k 10 200 python3 $R/tools/experiments/minres_iter_time.py 1000000 32 | tail -2
