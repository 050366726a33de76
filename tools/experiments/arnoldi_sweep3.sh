for pt in 64 128 160 320; do echo "== per_thread $pt"; HIPEIG_ARNOLDI_PER_THREAD=$pt python tools/experiments/arnoldi_bench.py 10000000 28 8 | tail -1; done
for pt in 12 16 20 24; do echo "== per_thread $pt 1e6"; HIPEIG_ARNOLDI_PER_THREAD=$pt python tools/experiments/arnoldi_bench.py 1000000 28 | tail -1; done
