#!/bin/bash
# Profile set of one round (run on the GPU box through gpurun):  gpurun --timeout 1200 -- 'bash tools/profile_round.sh r4a'
# then, here:  python tools/collect_profiles.py r4a r4 "PN15QP880 k=4"
# Every profiled pass runs the SAME command, `python3 bench.py --no-cpu --no-extras` with MKHE_NO_OVERLAP=1 (each kernel alone on the
# main stream).  With --no-extras the dominant kernel (the Decompose-fused forward NTT) is launched exactly 2 * (W + K) + 300 + K
# times: once per MulRelin since round 6 (the 1792-limb hoisting launch; the t_i are transformed inside ntt16_f2_kernel) -- in the cold-start
# leg (W + K), the 100 + 200 steps of the steady-state leg, the timed region (W + K) and the HIP-event leg (K).
# Round 6: `bash tools/profile_round.sh r6a part1`, `... part2`, `... part3` (no second argument = these three); then, here, tools/collect_profiles.py (-> traffic.json),
# then `... r6a part4` (the bench lines again WITH the traffic figures, the PMC passes of the round-5 launch set, the phase trace, N > 1) and collect once more.
TAG=${1:-r5}
PART=${2:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
P="rocprofv3 --output-format csv --kernel-trace"
if [ "$PART" = part1 ] || [ "$PART" = all ]; then
python3 bench.py > $O/bench_plain.json 2> $O/bench_plain.err
MKHE_NTT32=1 python3 bench.py --no-cpu > $O/bench_plain_h32.json 2> $O/bench_plain_h32.err
MKHE_NTT32=0 python3 bench.py --no-cpu > $O/bench_plain_h16.json 2> $O/bench_plain_h16.err
echo "plain done" 
MKHE_NO_OVERLAP=1 $P --stats -d $O/stats_noovl -o p -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-extras > $O/bench_noovl.json 2> $O/bench_noovl.err
$P --stats -d $O/stats_ovl -o p -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-extras > $O/bench_ovl.json 2> $O/bench_ovl.err
echo "stats done"
export MKHE_NO_OVERLAP=1
# (the counter passes run twice, the forward NTT kernel forced: the two-pass kernel (MKHE_NTT32=0), then the single-pass kernel (=1) -- figures for either at hand)
export MKHE_NTT32=0
$P --pmc FETCH_SIZE -d $O/pmc_fetch -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu --no-extras > $O/bench_pmc_fetch.json 2> $O/bench_pmc_fetch.err
$P --pmc WRITE_SIZE -d $O/pmc_write -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu --no-extras > $O/bench_pmc_write.json 2> $O/bench_pmc_write.err
export MKHE_NTT32=1
$P --pmc FETCH_SIZE -d $O/pmc_fetch_h32 -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu --no-extras > $O/bench_pmc_fetch_h32.json 2> $O/bench_pmc_fetch_h32.err
$P --pmc WRITE_SIZE -d $O/pmc_write_h32 -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu --no-extras > $O/bench_pmc_write_h32.json 2> $O/bench_pmc_write_h32.err
unset MKHE_NTT32
echo "traffic done"
$P --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d $O/sq_a -o p -- python3 bench.py --steps 4 --warmup 2 --no-cpu --no-extras > $O/sq_a.json 2> $O/sq_a.err
$P --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA -d $O/sq_b -o p -- python3 bench.py --steps 4 --warmup 2 --no-cpu --no-extras > $O/sq_b.json 2> $O/sq_b.err
$P --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d $O/sq_d -o p -- python3 bench.py --steps 4 --warmup 2 --no-cpu --no-extras > $O/sq_d.json 2> $O/sq_d.err
echo "sq done"
fi
if [ "$PART" = part2 ] || [ "$PART" = all ]; then
export MKHE_NO_OVERLAP=1
$P --stats -d $O/stats_bfv -o p -- python3 bench.py --scheme bfv --steps 10 --warmup 2 --no-cpu > $O/bench_bfv.json 2> $O/bench_bfv.err
unset MKHE_NO_OVERLAP
python3 bench.py --scheme bfv --steps 10 --warmup 2 > $O/bench_bfv_plain.json 2> $O/bench_bfv_plain.err
echo "bfv done"
MKHE_NO_OVERLAP=1 $P --stats -d $O/stats_pn16 -o p -- python3 bench.py --params PN16QP1761 --parties 8 --steps 6 --warmup 2 --no-cpu --no-extras > $O/bench_pn16_noovl.json 2> $O/bench_pn16_noovl.err
python3 bench.py --params PN16QP1761 --parties 8 --steps 10 --warmup 2 > $O/bench_pn16.json 2> $O/bench_pn16.err
echo "pn16 done"
python3 bench.py --scheme cnn --parties 2 --steps 20 --warmup 3 > $O/bench_cnn2.json 2> $O/bench_cnn2.err
python3 bench.py --scheme cnn --parties 4 --steps 20 --warmup 3 > $O/bench_cnn4.json 2> $O/bench_cnn4.err
python3 bench.py --params PN14QP439 --steps 20 --warmup 3 > $O/bench_pn14.json 2> $O/bench_pn14.err
python3 bench.py --parties 2 > $O/bench_2party.json 2> $O/bench_2party.err
python3 bench.py --scheme cnn --parties 4 --steps 20 --warmup 3 --batch 8 > $O/bench_cnn4_batch8.json 2> $O/bench_cnn4_batch8.err
python3 bench.py --scheme cnn --parties 4 --steps 20 --warmup 3 --no-cpu --batch 16 > $O/bench_cnn4_batch16.json 2> $O/bench_cnn4_batch16.err
python3 bench.py --scheme cnn --parties 2 --steps 20 --warmup 3 --no-cpu --batch 8 > $O/bench_cnn2_batch8.json 2> $O/bench_cnn2_batch8.err
for B in 4 8 16; do python3 bench.py --params PN14QP439 --steps 30 --warmup 3 --no-cpu --no-extras --batch $B 2>/dev/null; done > $O/pn14_batch.jsonl
for k in 1 2 4 8 16; do python3 bench.py --parties $k --no-cpu --device-keys --steps 20 --warmup 3 2>/dev/null; done > $O/party_sweep.jsonl
echo "secondary done"
fi
if [ "$PART" = part3 ] || [ "$PART" = all ]; then
# issue-rate microbenchmarks (built from source here: no binaries in the tree), power / clock under the dominant kernel, steady-state ablation table of the shipped kernel
for u in bfly30u_rate bfly31_rate valu_rate bfly_asm_rate read_bw; do [ -f tools/ubench/$u.hip ] && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/ubench/$u tools/ubench/$u.hip 2>/dev/null; done
(echo "== tools/ubench/bfly30u_rate.hip"; tools/ubench/bfly30u_rate; echo "== tools/ubench/bfly31_rate.hip"; tools/ubench/bfly31_rate; echo "== tools/ubench/valu_rate.hip"; tools/ubench/valu_rate; echo "== tools/ubench/bfly_asm_rate.hip"; tools/ubench/bfly_asm_rate; echo "== tools/ubench/read_bw.hip"; tools/ubench/read_bw) > $O/ubench.txt 2>&1
bash tools/power_probe.sh > $O/power_probe.txt 2>&1
# steady-state table of the single-pass kernel (H32: the default for launches of 512 limbs and more) and of the two-pass kernel it replaced there
(echo "## ntt32_fwd_kernel (MKHE_NTT32=1)"; export MKHE_NTT32=1; SRC=ntt32_kernels REPS=1500 bash tools/ntt16_variants.sh "shipped:" "no_butterflies:-DMKHE_H32_X_NOBFLY" "no_exchanges:-DMKHE_H32_X_NOXCHG" "no_stores:-DMKHE_H32_X_NOSTORE" \
    "no_source_loads:-DMKHE_H16_X_NOSRC" "no_twiddle_loads:-DMKHE_H16_X_NOTWLOAD -DMKHE_H32_X_NOTWB" "butterflies_only:-DMKHE_H32_X_NOXCHG -DMKHE_H32_X_NOSTORE -DMKHE_H16_X_NOSRC -DMKHE_H16_X_NOTWLOAD -DMKHE_H32_X_NOTWB" \
    "no_butterflies_no_twiddles:-DMKHE_H32_X_NOBFLY -DMKHE_H16_X_NOTWLOAD -DMKHE_H32_X_NOTWB" "one_butterfly_per_asm_block:-DMKHE_H32_BF2=0" "no_phase_priorities:-DMKHE_H32_PHPRIO=0" "one_priority_set:-DMKHE_H32_PHPRIO=1" \
    "ring_8:-DMKHE_H32_RING=8" "prefetch_next_limb:-DMKHE_H32_PREFETCH=1" "round3_schedule_byte_load:-DMKHE_H16_X_SCHEDBYTE" "shipped_again:"
 export MKHE_NTT32=0; echo "## ntt16_fwd_kernel<true> (MKHE_NTT32=0)"; REPS=1500 bash tools/ntt16_variants.sh "shipped:" "no_mem_no_xchg:-DMKHE_H16_X_NOTWLOAD -DMKHE_H16_X_NOSRC -DMKHE_H16_X_NOSTORE -DMKHE_H16_X_NOXCHG=15" \
    "no_bfly:-DMKHE_H16_X_NOBFLY=3" "round3_schedule_byte_load:-DMKHE_H16_X_SCHEDBYTE" "shipped_again:"; unset MKHE_NTT32) > $O/ntt16_ablation.txt 2>&1
# the fused Decompose + step-F2 kernel (round 6) inside the MulRelin: shipped, its experiments, and its timing-only ablations
bash tools/f2_variants.sh "shipped:" "unpipelined_source_loads:-DMKHE_F2_SPIPE=0" "phase_C_twiddles_from_L2:-DMKHE_F2_TWLDS=0" "no_wave_priorities:-DMKHE_F2_PHPRIO=0" "key_ring_4_pairs_12:-DMKHE_F2_KRING=4 -DMKHE_F2_SP=12" \
    "no_products_no_keys:-DMKHE_F2_X_NOMAC -DMKHE_F2_X_NOKEYS" "no_key_loads:-DMKHE_F2_X_NOKEYS" "no_source_loads:-DMKHE_H16_X_NOSRC" "no_exchanges:-DMKHE_H16_X_NOXCHG=15" "no_twiddle_loads:-DMKHE_H16_X_NOTWLOAD" \
    "vector_alu_only:-DMKHE_H16_X_NOTWLOAD -DMKHE_H16_X_NOSRC -DMKHE_H16_X_NOXCHG=15 -DMKHE_F2_X_NOKEYS" "shipped_again:" > $O/f2_variants.txt 2>&1
(echo "== fused (product library)"; python3 bench.py --no-cpu --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'MulRelin/s', {k.split()[0]: round(v['ms_per_step']*1e3,1) for k,v in d['roofline']['kernels'].items()})"
 for i in 1 2 3; do for m in 0 1; do echo "== MKHE_F2_FUSED=$m (switches library)"; MKHE_LIB=$R/mkhe-kklss_amd/lib/libmkhe_hip_switches.so MKHE_F2_FUSED=$m python3 bench.py --no-cpu --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'MulRelin/s', round(d['config']['mulrelin_per_sec_cold_start'],1), 'cold', {k.split()[0]: round(v['ms_per_step']*1e3,1) for k,v in d['roofline']['kernels'].items()})"; done; done) > $O/f2_fused_ab.txt 2>&1
(echo "== MKHE_NTT32=0 (ntt16_fwd_kernel<true>)"; MKHE_NTT32=0 python3 tools/ntt16_bench.py 1500; echo "== MKHE_NTT32=1 (ntt32_fwd_kernel)"; MKHE_NTT32=1 python3 tools/ntt16_bench.py 1500) > $O/ntt16_bench.txt 2>&1
# the Decompose launches INSIDE the MulRelin (kernel trace of the bench command, second half of the run): the two launch sizes apart, both kernels
bash tools/trace_ntt_in_context.sh h16:MKHE_NTT32=0 h32:MKHE_NTT32=1 h16_again:MKHE_NTT32=0 h32_again:MKHE_NTT32=1 > $O/ntt_in_context.txt 2>&1
rm -rf $R/gpurun_out/ctx
echo "ntt done"
fi
if [ "$PART" = part4 ]; then
# AFTER tools/collect_profiles.py has written profiles/traffic.json for these sources: the bench lines again, so that the committed lines carry
# roofline.traffic too (round 5's were recorded before and say null); the PMC passes of the round-5 launch set on the same box; the phase trace; N > 1
python3 bench.py > $O/bench_plain.json 2> $O/bench_plain.err
MKHE_NTT32=1 python3 bench.py --no-cpu > $O/bench_plain_h32.json 2> $O/bench_plain_h32.err
MKHE_NTT32=0 python3 bench.py --no-cpu > $O/bench_plain_h16.json 2> $O/bench_plain_h16.err
echo "lines done"
export MKHE_NO_OVERLAP=1 MKHE_NTT32=0 MKHE_LIB=$R/mkhe-kklss_amd/lib/libmkhe_hip_switches.so MKHE_F2_FUSED=0
$P --pmc FETCH_SIZE -d $O/pmc_fetch_unfused -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu --no-extras > /dev/null 2> $O/pmc_fetch_unfused.err
$P --pmc WRITE_SIZE -d $O/pmc_write_unfused -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu --no-extras > /dev/null 2> $O/pmc_write_unfused.err
unset MKHE_NO_OVERLAP MKHE_NTT32 MKHE_LIB MKHE_F2_FUSED
(echo "# HBM bytes per launch, (2 FETCH_SIZE + WRITE_SIZE) 1024, of the ROUND-5 launch set (MKHE_F2_FUSED=0, diagnostic library, MKHE_NTT32=0) on the box of this round's traffic.json"
 python3 tools/traffic_from_pmc.py $O/pmc_fetch_unfused $O/pmc_write_unfused "PN15QP880 k=4 (MKHE_F2_FUSED=0)" 6 2 | python3 -c "
import json, sys
d = json.load(sys.stdin); tot = 0.0
per_step = {}
for k, v in d['kernels'].items():
    if '<' not in k and any(x.startswith(k + '<') for x in d['kernels']): continue          # (the un-templated alias of a templated name)
    n = v['launches'] / 322.0
    per_step[k] = (n, v['hbm_bytes_per_launch'])
    tot += n * v['hbm_bytes_per_launch']
for k, (n, b) in sorted(per_step.items(), key=lambda kv: -kv[1][0] * kv[1][1]): print('%-34s %4.1f launches / step  %8.1f MB / launch' % (k, n, b / 1e6))
print('per step: %.2f GB' % (tot / 1e9))") > $O/pmc_unfused.txt 2>&1
echo "unfused pmc done"
[ -f $R/mkhe-kklss_amd/lib/libmkhe_hip_trace.so ] && MKHE_LIB=$R/mkhe-kklss_amd/lib/libmkhe_hip_trace.so timeout -k 10 300 python3 tools/f2_trace.py > $O/f2_trace.txt 2>&1
MKHE_DIST_BACKEND=gloo MKHE_DIST_ONE_DEVICE=1 timeout -k 10 600 python3 bench.py --gpus 5 --steps 2 --warmup 1 > $O/dist_5ranks.json 2> $O/dist_5ranks.err
echo "part4 done"
fi
find $O -name '*kernel_trace.csv' -path '*stats_*' -delete
find $O -name '*agent_info.csv' -delete
du -sh $O
