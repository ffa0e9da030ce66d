#!/bin/bash
# SQ counter passes over bench.py (no overlap): where do the NTT kernels' wave cycles go?
TAG=${1:-sq}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
export MKHE_NO_OVERLAP=1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d $O/a -o p -- python3 bench.py --steps 4 --warmup 2 --no-cpu > $O/a.json 2> $O/a.err
rocprofv3 --output-format csv --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA -d $O/b -o p -- python3 bench.py --steps 4 --warmup 2 --no-cpu > $O/b.json 2> $O/b.err
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 -d $O/c -o p -- python3 bench.py --steps 4 --warmup 2 --no-cpu > $O/c.json 2> $O/c.err
tail -3 $O/a.err $O/b.err $O/c.err
ls -la $O/*
