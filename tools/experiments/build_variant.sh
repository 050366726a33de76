#!/bin/bash
# build_variant.sh NAME "-DFLAG=1 ..."  ->  eigensolvers_amd/libhipeig_NAME.so (for A/B runs via HIPEIG_LIB)
set -e
NAME=$1; FLAGS=$2
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
D=$(mktemp -d)
cd "$ROOT/eigensolvers_amd/csrc"
for f in ctx comm comm_direct blas1 spmv spmm generate minres minres_block dense_small; do
  /opt/rocm/bin/hipcc $FLAGS -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -c $f.hip -o $D/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhipeig_$NAME.so $D/*.o -ldl
rm -rf $D
echo built libhipeig_$NAME.so
