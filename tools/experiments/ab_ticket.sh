for lib in libhipeig.so libhipeig_tk0.so libhipeig.so libhipeig_tk0.so; do
  echo "== $lib"; HIPEIG_LIB=$PWD/eigensolvers_amd/$lib timeout -k 10 200 python tools/experiments/minres_iter_time.py 1000000 32 2>&1 | tail -4
done
for lib in libhipeig.so libhipeig_tk0.so; do
  echo "== $lib 1e7"; HIPEIG_LIB=$PWD/eigensolvers_amd/$lib timeout -k 10 300 python tools/experiments/minres_iter_time.py 10000000 64 2>&1 | tail -4
done
