#!/bin/bash
# Builds variants of csrc/ntt16_kernels.hip (compile-time switches) against the objects of the normal build and times the 1792 / 896-limb
# Decompose launches back to back (tools/ntt16_bench.py; digests show whether a variant still computes the same hoisted digits).
#   gpurun -- 'bash tools/ntt16_variants.sh "name1:-DFLAG1 -DFLAG2" "name2:..." > gpurun_out/variants.txt 2>&1'
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/mkhe-kklss_amd/csrc
B=$R/mkhe-kklss_amd/build
SRC=${SRC:-ntt16_kernels}        # the source the variants re-compile (SRC=ntt32_kernels: the single-pass kernel)
make -s -C $C -j8 product > /dev/null 2>&1
run() {
    name=$1; shift
    mkdir -p $B/var_$name
    cp $B/*.o $B/var_$name/      # every object of the product build (round 5 copied a stale list: the variant libraries did not link and the table came out empty)
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DMKHE_ABLATION -I$C $* -c $C/$SRC.hip -o $B/var_$name/$SRC.o 2>/dev/null || { echo "$name: build failed"; return; }
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/var_$name/lib.so $B/var_$name/*.o || { echo "$name: link failed"; return; }
    echo "== $name   ($*)"
    MKHE_LIB=$B/var_$name/lib.so python3 $R/tools/ntt16_bench.py ${REPS:-10} 2>&1 | grep -E "limbs +(1792|896) " | cut -c1-150
    rm -rf $B/var_$name
}
for v in "$@"; do
    name=${v%%:*}; flags=${v#*:}
    [ "$flags" = "$v" ] && flags=""
    run $name $flags
done
