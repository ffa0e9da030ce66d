#!/usr/bin/env python3
"""Static check of the hand-issued loads of csrc/ntt16_kernels.hip (ld_issue + counted s_waitcnt): walks the gfx950 ISA of every kernel in
program order with the in-order completion model of vmcnt and reports any instruction that touches the destination registers of a vector
load that may still be in flight (a register spill placed by the allocator between an asm load and its wait reads garbage -- the compiler
does not know that the asm statement is a load).  Straight-line model: branches are ignored, every label resets nothing.
  python tools/check_inflight.py [extra hipcc flags]     -> exit code 1 if anything is flagged"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "mkhe-kklss_amd", "csrc", os.environ.get("CHECK_SRC", "ntt16_kernels.hip"))
asm = "/tmp/check_inflight.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-S",
                       "--cuda-device-only", src, "-o", asm] + sys.argv[1:], stderr=subprocess.DEVNULL)
def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(3) is not None: out.add(int(m.group(3)))
        else: out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out
bad = 0
kernel = None
for ln in open(asm):
    t = ln.strip()
    m = re.match(r"^(_ZN4mkhe3h16\w+):", t)
    if m:
        kernel = m.group(1); inflight = []; continue          # (kind, dest regs)
    if kernel is None or not t or t[0] in ";." or t.endswith(":"):
        continue
    if t.startswith("s_endpgm"):
        kernel = None; continue
    op = t.split()[0]
    body = t.split(";")[0]
    if op.startswith(("global_load", "scratch_load", "buffer_load")):
        ops = body[len(op):].split(",")
        # address / data registers of the load itself must not be in flight either
        used = regs(",".join(ops[1:]))
        for k, d in inflight:
            if k == "ld" and d & used:
                print("%s: %s   <- address uses a register of a load in flight" % (kernel[-40:], t)); bad += 1
        inflight.append(("ld", regs(ops[0])))
        continue
    if op.startswith(("global_store", "scratch_store", "buffer_store", "global_atomic")):
        used = regs(body[len(op):])
        for k, d in inflight:
            if k == "ld" and d & used:
                print("%s: %s   <- stores a register of a load in flight" % (kernel[-40:], t)); bad += 1
        inflight.append(("st", set()))
        continue
    if op == "s_waitcnt":
        m = re.search(r"vmcnt\((\d+)\)", body)
        if m:
            n = int(m.group(1))
            inflight = inflight[len(inflight) - n:] if n and n < len(inflight) else ([] if n == 0 else inflight)
        continue
    used = regs(body[len(op):])
    for k, d in inflight:
        if k == "ld" and d & used:
            print("%s: %s   <- touches a register of a load in flight" % (kernel[-40:], t)); bad += 1
            break
print("check_inflight: %d problem(s)" % bad)
sys.exit(1 if bad else 0)
