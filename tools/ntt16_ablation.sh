#!/bin/bash
# Ablation builds of the dominant kernel (csrc/ntt16_kernels.hip): what is left of a Decompose launch when the butterflies, the LDS
# exchanges, the result stores or the parked half are taken out (WRONG RESULTS on purpose, timing only).  One gpurun call:
#   gpurun -- 'bash tools/ntt16_ablation.sh > gpurun_out/ntt16_ablation.txt 2>&1'
# Each variant re-compiles ntt16_kernels.hip with the switch, links it with the objects of the normal build and runs tools/ntt16_bench.py
# (launches of 1792 / 896 limbs back to back: the chip runs them at its throttled clock, about 0.20 us per limb against 0.15 us inside a MulRelin).
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/mkhe-kklss_amd/csrc
B=$R/mkhe-kklss_amd/build
make -s -C $C -j8 > /dev/null 2>&1
run() {
    name=$1; shift
    mkdir -p $B/abl_$name
    for f in ntt_kernels poly_kernels keygen_kernels engine keygen capi; do cp $B/$f.o $B/abl_$name/$f.o; done
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DMKHE_ABLATION -I$C "$@" -c $C/ntt16_kernels.hip -o $B/abl_$name/ntt16_kernels.o 2>/dev/null || { echo "$name: build failed"; return; }
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/abl_$name/lib.so $B/abl_$name/*.o
    echo "== $name   ($*)"
    MKHE_LIB=$B/abl_$name/lib.so python3 $R/tools/ntt16_bench.py 10 2>&1 | grep -E "limbs +(1792|896) " | cut -c1-112
    rm -rf $B/abl_$name
}
run shipped
run no_scalar_butterflies   -DMKHE_H16_X_NOBFLY=1
run no_perlane_butterflies  -DMKHE_H16_X_NOBFLY=2
run no_butterflies          -DMKHE_H16_X_NOBFLY=3
run no_butterflies_no_xchg  -DMKHE_H16_X_NOBFLY=3 -DMKHE_H16_X_NOXCHG=14
run no_butterflies_no_store -DMKHE_H16_X_NOBFLY=3 -DMKHE_H16_X_NOSTORE
run no_butterflies_no_twiddle_loads -DMKHE_H16_X_NOBFLY=3 -DMKHE_H16_X_NOTWLOAD
run no_butterflies_no_source_loads  -DMKHE_H16_X_NOBFLY=3 -DMKHE_H16_X_NOSRC
run skeleton_only           -DMKHE_H16_X_NOBFLY=3 -DMKHE_H16_X_NOTWLOAD -DMKHE_H16_X_NOSRC -DMKHE_H16_X_NOSTORE
run no_stash                -DMKHE_H16_NO_STASH
run shipped
