#!/bin/bash
# A/B of the planned grid of ntt16_f2_kernel (MKHE_F2_FUSED=1, the default, against 2 = one workgroup per CU or nothing; diagnostic library): shapes whose passes do not fill the chip twice -- one to three parties, low
# levels -- against the unfused launch set they ran on before.  Same call, same box, alternating.
#   gpurun -- 'bash tools/f2_plan_ab.sh > gpurun_out/f2_plan_ab.txt 2>&1'
R=${GRAFT_REPO_ROOT:-$(pwd)}
export MKHE_LIB=$R/mkhe-kklss_amd/lib/libmkhe_hip_switches.so
line() { python3 - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
c = d["config"]
ks = d["roofline"]["kernels"]
def us(name):
    return sum(v["ms_per_step"] * 1000 for k, v in ks.items() if k.startswith(name))
print("   %8.1f %s  (cold %.1f)  rotate %s conjugate %s | Decompose NTT %.1f  f2 %.1f  products %.1f  inverse NTT %.1f  ModDown %.1f us/step" % (
    d["value"], d["unit"], c.get("mulrelin_per_sec_cold_start", 0), c.get("rotate_per_sec"), c.get("conjugate_per_sec"),
    us("ntt16_fwd_kernel<true>") + us("ntt32_fwd_kernel<true>"), us("ntt16_f2_kernel"), us("ext_inner_kernel"), us("ntt_inv_kernel"), us("moddown")))
PY
}
for k in ${PARTIES:-1 2 3 4}; do
    for rep in 1 2; do
        for plan in 1 0; do
            echo "== parties $k  MKHE_F2_FUSED=$((2 - plan))"
            MKHE_F2_FUSED=$((2 - plan)) python3 $R/bench.py --parties $k --no-cpu --steps 40 --warmup 5 > /tmp/ab_${k}_${plan}.json 2>/tmp/ab.err || { tail -3 /tmp/ab.err; exit 1; }
            line /tmp/ab_${k}_${plan}.json
        done
    done
done
echo "== mkbfv 2 parties"
for plan in 1 0 1 0; do
    MKHE_F2_FUSED=$((2 - plan)) python3 $R/bench.py --scheme bfv --parties 2 --no-cpu --no-extras --steps 30 --warmup 5 > /tmp/ab_bfv.json 2>/tmp/ab.err || { tail -3 /tmp/ab.err; exit 1; }
    echo "   MKHE_F2_FUSED=$((2 - plan))"; line /tmp/ab_bfv.json
done
echo "== cnn 4 parties / 2 parties"
for plan in 1 0 1 0; do
    for k in 4 2; do
        MKHE_F2_FUSED=$((2 - plan)) python3 $R/bench.py --scheme cnn --parties $k --no-cpu --no-extras --steps 20 --warmup 3 > /tmp/ab_cnn.json 2>/tmp/ab.err || { tail -3 /tmp/ab.err; exit 1; }
        python3 -c "
import json,sys
d=json.loads([l for l in open('/tmp/ab_cnn.json') if l.startswith('{')][-1]); print('   MKHE_F2_FUSED=$((2 - plan)) parties $k: %.1f %s (%.3f ms)' % (d['value'], d['unit'], d['ms_per_step']))"
    done
done
