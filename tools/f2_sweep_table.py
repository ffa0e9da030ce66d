"""Joins the outputs of tools/f2_level_sweep.py (planned / unfused, alternating) into the table of profiles/r6_f2_plan_sweep.txt:
python3 tools/f2_sweep_table.py planned1.txt unfused1.txt planned2.txt unfused2.txt"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_f2_schedule import _probe
def rd(f):
    d = {}
    for l in open(f):
        if l.startswith('#'): continue
        k, L, v = l.split(); d[(int(k), int(L))] = float(v)
    return d
a1, b1, a2, b2 = [rd(f) for f in sys.argv[1:5]]
print("# MulRelinNew per second on PN15QP880 (tools/f2_level_sweep.py, one box, four processes in this order: planned, MKHE_F2_FUSED=0, planned, MKHE_F2_FUSED=0)")
print("# plan = workgroups/parts of f2_plan_schedule for a 256-CU device (0/0: no fused launch for the shape: both columns run the same launches, their difference is the noise of the method)")
print("parties level | planned  unfused  planned  unfused | gain1  gain2 | plan")
for key in sorted(a1):
    k, L = key
    n, p, _ = _probe(k, L + 1, L + 3, [100] * (L + 3), -256)
    print("%d %2d | %8.1f %8.1f %8.1f %8.1f | %+5.1f%% %+5.1f%% | %d/%d" % (k, L, a1[key], b1[key], a2[key], b2[key], 100 * (a1[key] / b1[key] - 1), 100 * (a2[key] / b2[key] - 1), n, p))
