#!/bin/bash
# Builds ONE variant library here (hipcc cross-compiles gfx950 without a GPU): tools/build_variant.sh <name> <source> [flags...]
# -> mkhe-kklss_amd/build/var_<name>/lib.so = every object of the normal product build + <source>.hip re-compiled with -DMKHE_ABLATION and the flags;
# run with MKHE_LIB=that file (wrong results on purpose when a MKHE_*_X_* switch is among the flags: timing only).
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/mkhe-kklss_amd/csrc
B=$R/mkhe-kklss_amd/build
name=$1; src=$2; shift 2
make -s -C $C -j8 product > /dev/null 2>&1 || { echo "normal build failed"; exit 1; }
mkdir -p $B/var_$name
cp $B/*.o $B/var_$name/
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DMKHE_ABLATION -I$C "$@" -c $C/$src.hip -o $B/var_$name/$src.o 2> $B/var_$name/build.log || { echo "$name: build failed"; tail -5 $B/var_$name/build.log; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/var_$name/lib.so $B/var_$name/*.o && rm -f $B/var_$name/*.o && echo "built $B/var_$name/lib.so"
