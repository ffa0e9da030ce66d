#!/bin/bash
# On the GPU box: times tools/ntt16_bench.py for every prebuilt variant library named on the command line (tools/build_variant.sh builds them here).
#   gpurun -- 'bash tools/run_variants.sh "name[:ENV=VAL,ENV2=VAL2]" ... > gpurun_out/variants.txt 2>&1'      (name "shipped" = the normal library)
R=${GRAFT_REPO_ROOT:-$(pwd)}
B=$R/mkhe-kklss_amd/build
for v in "$@"; do
    name=${v%%:*}; envs=${v#*:}; [ "$envs" = "$v" ] && envs=""
    lib=$B/var_$name/lib.so; [ "$name" = shipped ] && lib=$R/mkhe-kklss_amd/lib/libmkhe_hip.so
    echo "== $name   ($envs)"
    ( for e in ${envs//,/ }; do export "$e"; done; MKHE_LIB=$lib timeout -k 10 240 python3 $R/tools/ntt16_bench.py ${REPS:-300} 2>&1 | grep -E "limbs +(1792|896) " | cut -c1-150 ) || echo "   (failed or timed out)"
done
