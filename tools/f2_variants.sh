#!/bin/bash
# Builds variants of csrc/ntt16_f2_kernels.hip (compile-time switches; the MKHE_*_X_* ones give WRONG results on purpose: timing only) against the objects
# of the product build and runs the headline bench (no CPU leg, no extras) with each: MulRelin/s and the fused kernel's time per launch from the HIP-event leg.
#   gpurun -- 'bash tools/f2_variants.sh "name1:-DFLAG1 -DFLAG2" "name2:..." > gpurun_out/f2_variants.txt 2>&1'
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/mkhe-kklss_amd/csrc
B=$R/mkhe-kklss_amd/build
make -s -C $C -j8 product > /dev/null 2>&1
run() {
    name=$1; shift
    mkdir -p $B/var_$name
    cp $B/*.o $B/var_$name/
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DMKHE_ABLATION -I$C $* -c $C/ntt16_f2_kernels.hip -o $B/var_$name/ntt16_f2_kernels.o 2>/dev/null || { echo "$name: build failed"; return; }
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/var_$name/lib.so $B/var_$name/*.o || { echo "$name: link failed"; return; }
    MKHE_LIB=$B/var_$name/lib.so timeout -k 10 300 python3 $R/bench.py --no-cpu --no-extras > $B/var_$name/b.json 2>/dev/null
    python3 - $B/var_$name/b.json "$name" "$*" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k = d["roofline"]["kernels"]
    f2 = [v for n, v in k.items() if n.startswith("ntt16_f2_kernel")]
    print("== %-28s %7.1f MulRelin/s   ntt16_f2_kernel %6.1f us/launch   (%s)" % (sys.argv[2], d["value"], f2[0]["avg_launch_us"] if f2 else float("nan"), sys.argv[3]))
except Exception as e:
    print("== %s: FAILED (%s)" % (sys.argv[2], e))
PY
    rm -rf $B/var_$name
}
for v in "$@"; do
    name=${v%%:*}; flags=${v#*:}
    [ "$flags" = "$v" ] && flags=""
    run $name $flags
done
