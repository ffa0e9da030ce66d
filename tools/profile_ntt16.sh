#!/bin/bash
# SQ / TCP / TCC counter passes over tools/ntt16_bench.py (Decompose launches of 1792 / 896 / 448 limbs at N = 2^15)
TAG=${1:-ntt16sq}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
P="rocprofv3 --output-format csv --kernel-trace"
$P --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d $O/a -o p -- python3 tools/ntt16_bench.py 6 > $O/a.log 2> $O/a.err
$P --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA -d $O/b -o p -- python3 tools/ntt16_bench.py 6 > $O/b.log 2> $O/b.err
$P --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 -d $O/c -o p -- python3 tools/ntt16_bench.py 6 > $O/c.log 2> $O/c.err
$P --pmc FETCH_SIZE -d $O/d -o p -- python3 tools/ntt16_bench.py 6 > $O/d.log 2> $O/d.err
$P --pmc WRITE_SIZE -d $O/e -o p -- python3 tools/ntt16_bench.py 6 > $O/e.log 2> $O/e.err
$P --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/f -o p -- python3 tools/ntt16_bench.py 6 > $O/f.log 2> $O/f.err
$P --pmc GRBM_GUI_ACTIVE SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS -d $O/g -o p -- python3 tools/ntt16_bench.py 6 > $O/g.log 2> $O/g.err
tail -2 $O/*.err
python3 tools/pmc_summary.py $O/a $O/b $O/c $O/d $O/e $O/f $O/g > $O/summary.txt 2>&1
cat $O/summary.txt
