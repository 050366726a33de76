#!/bin/bash
# window / bin size of the pair layout at N = 1e7 (tools/pair_bench.py); output: gpurun_out/pair_sweep.txt
mkdir -p gpurun_out
: > gpurun_out/pair_sweep.txt
for wb in 17 16 15; do for bb in 5 4 3; do
  echo "wbits $wb binbits $bb" >> gpurun_out/pair_sweep.txt
  HIPEIG_TCOOW_PAIR_WBITS=$wb HIPEIG_TCOOW_PAIR_BINBITS=$bb timeout -k 10 120 python tools/pair_bench.py ${1:-10000000} >> gpurun_out/pair_sweep.txt 2>> gpurun_out/pair_sweep.err || exit 1
done; done
