#!/bin/bash
# Round 3: ablations of the SHIPPED kernel with the butterflies KEPT (wrong results on purpose, timing only): what one memory / LDS
# component costs the full kernel (exposed latency), as opposed to tools/ntt16_ablation.sh, which removes the butterflies first.
#   gpurun -- 'bash tools/ntt16_ablation_full.sh > gpurun_out/ntt16_ablation_full.txt 2>&1'
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
run full_no_twiddle_loads -DMKHE_H16_X_NOTWLOAD
run full_no_source_loads  -DMKHE_H16_X_NOSRC
run full_no_store         -DMKHE_H16_X_NOSTORE
run full_no_xchg          -DMKHE_H16_X_NOXCHG=14
run full_no_xchg_all      -DMKHE_H16_X_NOXCHG=15
run full_no_mem           -DMKHE_H16_X_NOTWLOAD -DMKHE_H16_X_NOSRC -DMKHE_H16_X_NOSTORE
run full_no_mem_no_xchg   -DMKHE_H16_X_NOTWLOAD -DMKHE_H16_X_NOSRC -DMKHE_H16_X_NOSTORE -DMKHE_H16_X_NOXCHG=15
run shipped
