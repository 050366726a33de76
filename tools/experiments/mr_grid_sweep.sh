# KC / KD grids: elements per thread (4 = n/1024 workgroups ... 32 = n/8192), time per MINRES iteration
for pt in 4 8 16 24 32 64; do
  echo "== per_thread $pt"
  HIPEIG_MR_PER_THREAD=$pt python tools/experiments/minres_iter_time.py 1000000 32 2>&1 | grep "fuse_kd 1" | tail -1
  HIPEIG_MR_PER_THREAD=$pt python tools/experiments/minres_iter_time.py 10000000 64 2>&1 | grep "fuse_kd 1" | tail -1
done
