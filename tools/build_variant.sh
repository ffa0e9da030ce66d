#!/bin/bash
# Builds ONE variant library here (hipcc cross-compiles gfx950 without a GPU): tools/build_variant.sh <name> <source.hip> [flags...]
# -> mkhe-kklss_amd/build/var_<name>/lib.so (objects of the normal build + the re-compiled source); run with MKHE_LIB=that file.
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/mkhe-kklss_amd/csrc
B=$R/mkhe-kklss_amd/build
name=$1; src=$2; shift 2
make -s -C $C -j8 > /dev/null 2>&1 || { echo "normal build failed"; exit 1; }
mkdir -p $B/var_$name
for f in ntt_kernels ntt16_kernels ntt32_kernels poly_kernels keygen_kernels engine batch keygen capi; do cp $B/$f.o $B/var_$name/$f.o; done
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DMKHE_ABLATION -I$C "$@" -c $C/$src.hip -o $B/var_$name/$src.o 2> $B/var_$name/build.log || { echo "$name: build failed"; tail -5 $B/var_$name/build.log; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/var_$name/lib.so $B/var_$name/*.o && rm -f $B/var_$name/*.o && echo "built $B/var_$name/lib.so"
