#!/bin/bash
# The GPU parity suite under the A/B switch sets of DESIGN.md section 6 (run on the GPU box):  gpurun --timeout 1200 -- 'bash tools/switch_matrix.sh'
# Every path a measurement decided against -- and every H16 kernel on every launch shape, thresholds at 1 -- has to give the same bits as the defaults.
# Each set is one pytest process (the switches are read once per process); a GPU fault anywhere fails the set.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/switch_matrix
mkdir -p $O
cd $R
# the switches exist only in the -DMKHE_SWITCHES build of the same sources (csrc/switches.h; `make -C mkhe-kklss_amd/csrc switches`)
LIB=$R/mkhe-kklss_amd/lib/libmkhe_hip_switches.so
[ -f $LIB ] || { echo "missing $LIB"; exit 2; }
rc=0
run() {
    local name=$1; shift
    env MKHE_LIB=$LIB "$@" python3 -m pytest tests -m gpu -x -q > $O/$name.txt 2>&1
    local r=$?
    if grep -q "Memory access fault" $O/$name.txt; then r=99; fi
    echo "$name: rc=$r  $(tail -1 $O/$name.txt)"
    [ $r -ne 0 ] && rc=1
    return $r
}
# Round 6: seven sets (the switches whose alternative lost in two rounds or more are gone from the sources: DESIGN.md section 6).  About five minutes each:
# three gpurun calls -- `bash tools/switch_matrix.sh a`, then `... b`, then `... c`; no argument runs everything.
P=${1:-all}
if [ $P = a ] || [ $P = all ]; then
run defaults MKHE_UNUSED=1 &&
run thresholds_at_1 MKHE_NTT16_INV_MIN=1 MKHE_NTT16_MIN=1 MKHE_NTT14_MIN=1 MKHE_NTT14_INV_MIN=1 MKHE_NTT32=0 &&
run h32_everywhere MKHE_NTT32=1 MKHE_NTT32_MIN=1 MKHE_F2_BALANCE=5 || rc=1
fi
if { [ $P = b ] || [ $P = all ]; } && [ $rc -eq 0 ]; then
run fusions_off MKHE_F2_FUSED=0 MKHE_FUSE_E=0 MKHE_FUSE_RESCALE=0 MKHE_EXT_FUSED_MAX=0 MKHE_POOL_GB=1 &&
run classes_off MKHE_FUSE_Y=0 MKHE_SPREAD_RADIX4=0 MKHE_NTT16_RADIX4=0 MKHE_NTT16_INV=0 MKHE_H16_UCLASS=0 MKHE_H16_FCLASS=0 MKHE_NTT16_HALVES=0 MKHE_H16_SCHED=0 MKHE_F2_FUSED=2 || rc=1
fi
if { [ $P = c ] || [ $P = all ]; } && [ $rc -eq 0 ]; then
run round1_kernels MKHE_NTT32=0 MKHE_NTT16=0 MKHE_NTT16_INV=0 MKHE_NTT_LDS=0 MKHE_EXT_MERGE=0 MKHE_EXT_GROUP=0 MKHE_FUSE_X=0 MKHE_NO_OVERLAP=1 &&
run small_ring_fused_everywhere MKHE_EXT_FUSED_MAX=1000000 MKHE_EXT_FUSED_INV=0 MKHE_NTT16_HALVES=2 || rc=1
fi
exit $rc
