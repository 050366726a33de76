for pt in 4 8 16 32; do echo "== per_thread $pt"; HIPEIG_ARNOLDI_PER_THREAD=$pt python tools/experiments/arnoldi_bench.py 1000000 28 | tail -2; done
HIPEIG_ARNOLDI_PER_THREAD=8 python tools/experiments/arnoldi_bench.py 10000000 28 10 | tail -2
HIPEIG_ARNOLDI_PER_THREAD=16 python tools/experiments/arnoldi_bench.py 10000000 28 10 | tail -2
