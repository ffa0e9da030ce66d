#!/bin/bash
# Round-2 profile set (run on the GPU box through gpurun):  gpurun -- 'bash tools/profile_r2.sh r2'
# Every profiled pass runs the SAME command, `python3 bench.py --no-cpu --no-extras` with MKHE_NO_OVERLAP=1 (each kernel alone on the
# main stream): the dominant kernel (the Decompose-fused forward NTT) is then launched exactly 2 * (warmup + 2 * steps) times --
# twice per MulRelin, in the warm-up, the timed loop and the HIP-event leg -- and nothing else of its class runs.
TAG=${1:-r2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
P="rocprofv3 --output-format csv --kernel-trace"
python3 bench.py > $O/bench_plain.json 2> $O/bench_plain.err
MKHE_NO_OVERLAP=1 $P --stats -d $O/stats_noovl -o p -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-extras > $O/bench_noovl.json 2> $O/bench_noovl.err
$P --stats -d $O/stats_ovl -o p -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-extras > $O/bench_ovl.json 2> $O/bench_ovl.err
export MKHE_NO_OVERLAP=1
$P --pmc FETCH_SIZE -d $O/pmc_fetch -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu --no-extras > $O/bench_pmc_fetch.json 2> $O/bench_pmc_fetch.err
$P --pmc WRITE_SIZE -d $O/pmc_write -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu --no-extras > $O/bench_pmc_write.json 2> $O/bench_pmc_write.err
$P --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d $O/sq_a -o p -- python3 bench.py --steps 4 --warmup 2 --no-cpu --no-extras > $O/sq_a.json 2> $O/sq_a.err
$P --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA -d $O/sq_b -o p -- python3 bench.py --steps 4 --warmup 2 --no-cpu --no-extras > $O/sq_b.json 2> $O/sq_b.err
$P --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 -d $O/sq_c -o p -- python3 bench.py --steps 4 --warmup 2 --no-cpu --no-extras > $O/sq_c.json 2> $O/sq_c.err
$P --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d $O/sq_d -o p -- python3 bench.py --steps 4 --warmup 2 --no-cpu --no-extras > $O/sq_d.json 2> $O/sq_d.err
$P --stats -d $O/stats_bfv -o p -- python3 bench.py --scheme bfv --steps 10 --warmup 2 --no-cpu > $O/bench_bfv.json 2> $O/bench_bfv.err
unset MKHE_NO_OVERLAP
python3 bench.py --scheme bfv --steps 10 --warmup 2 > $O/bench_bfv_plain.json 2> $O/bench_bfv_plain.err
MKHE_NO_OVERLAP=1 $P --stats -d $O/stats_pn16 -o p -- python3 bench.py --params PN16QP1761 --parties 8 --steps 6 --warmup 2 > $O/bench_pn16_noovl.json 2> $O/bench_pn16_noovl.err
python3 bench.py --params PN16QP1761 --parties 8 --steps 10 --warmup 2 > $O/bench_pn16.json 2> $O/bench_pn16.err
python3 bench.py --scheme cnn --parties 2 --steps 20 --warmup 3 > $O/bench_cnn2.json 2> $O/bench_cnn2.err
python3 bench.py --scheme cnn --parties 4 --steps 20 --warmup 3 > $O/bench_cnn4.json 2> $O/bench_cnn4.err
python3 bench.py --params PN14QP439 --steps 20 --warmup 3 --no-cpu > $O/bench_pn14.json 2> $O/bench_pn14.err
for k in 1 2 4 8 16; do python3 bench.py --parties $k --no-cpu --device-keys --steps 20 --warmup 3 2>/dev/null; done > $O/party_sweep.jsonl
# issue-rate microbenchmarks and the per-phase timeline of the dominant kernel
(echo "== tools/ubench/bfly31_rate.hip"; tools/ubench/bfly31_rate; echo "== tools/ubench/bfly16_rate.hip"; tools/ubench/bfly16_rate; echo "== tools/ubench/valu_rate.hip"; tools/ubench/valu_rate; echo "== tools/ubench/bfly_rate.hip"; tools/ubench/bfly_rate) > $O/ubench.txt 2>&1
MKHE_LIB=$R/mkhe-kklss_amd/lib/libmkhe_hip_trace.so python3 tools/ntt16_trace.py 8 > $O/ntt16_trace.txt 2>&1
python3 tools/ntt16_bench.py > $O/ntt16_bench.txt 2>&1
find $O -name '*kernel_trace.csv' -path '*stats_*' -delete
find $O -name '*agent_info.csv' -delete
du -sh $O
