#!/usr/bin/env python3
"""Static check of the single-pass kernel's prefetch (MKHE_H32_PREFETCH): the 32 source words of the next limb are requested between the stores of
the current one and are in flight across the loop's back edge, invisible to the compiler -- so between a request and the counted wait that
covers it (s_waitcnt vmcnt(61 - 4 g) in stage 0) NO instruction may read or write the destination registers (a register-allocator copy or spill
there would move a value that has not landed).  Walks the gfx950 ISA of both kernels (hipcc -S, no GPU needed); exit code 1 on a violation."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "mkhe-kklss_amd", "csrc", "ntt32_kernels.hip")
asm = "/tmp/ntt32_inflight.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-S", "--cuda-device-only"] +
                      sys.argv[1:] + [src, "-o", asm], stderr=subprocess.DEVNULL)
txt = open(asm).read()
bad = 0
def regs_of(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()
def vregs(line):
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|v\d+", line.split(";")[0]):
        out |= regs_of(tok)
    return out
for kname in ("_ZN4mkhe3h3216ntt32_fwd_kernelILb1EEEvNS_8NttBatchE", "_ZN4mkhe3h3216ntt32_fwd_kernelILb0EEEvNS_8NttBatchE"):
    body = txt[txt.index(kname + ":"):]
    body = body[:body.index("s_endpgm")]
    lines = [l.strip() for l in body.split("\n")]
    labels = {l.split(":")[0]: i for i, l in enumerate(lines) if l.startswith(".LBB") and ":" in l}
    # tails: runs of `global_store ... nt` interleaved with global_load_dwordx2
    i, tails = 0, []
    while i < len(lines):
        if lines[i].startswith("global_store_dwordx2") and lines[i].endswith("nt"):
            j, loads, last = i, [], i
            while j < len(lines) and j - last < 40:
                if lines[j].startswith("global_store_dwordx2") and lines[j].endswith("nt"):
                    last = j
                elif lines[j].startswith("global_load_dwordx2"):
                    loads.append((j, regs_of(lines[j].split()[1].rstrip(","))))
                    last = j
                j += 1
            if len(loads) == 32:
                tails.append(loads)
            i = last + 1
        else:
            i += 1
    print(kname[14:38], "tails with 32 interleaved loads:", len(tails))
    if len(tails) != 2:
        print("   expected two (one per loop)"); bad += 1
    for loads in tails:
        inflight = set().union(*[r for _, r in loads])
        pair_regs = [loads[2 * g][1] | loads[2 * g + 1][1] for g in range(16)]
        # walk forward from the last load -- unconditional branches followed, conditional ones (loop exits, the cases of the job decode) not taken:
        # the fall-through path is the one that reaches the next limb -- until every pair has been released by its counted wait
        pos, steps, first_wait, released = loads[-1][0] + 1, 0, None, 0
        touched = []
        while steps < 6000 and inflight:
            l = lines[pos]
            steps += 1
            m = re.match(r"s_waitcnt vmcnt\((\d+)\)", l)
            if m:
                k = int(m.group(1))
                if first_wait is None:
                    first_wait = k
                if k == 0:
                    inflight = set(); break
                while released < 16 and 61 - 4 * released >= k:      # vmcnt(k) covers every pair whose own count is >= k
                    inflight -= pair_regs[released]; released += 1
            elif l.startswith("s_branch"):
                pos = labels[l.split()[1]]; continue
            elif l and not l.startswith((";", ".")):
                if l.startswith(("v_", "scratch_", "ds_", "global_", "buffer_", "flat_")) and vregs(l) & inflight:
                    touched.append(l)
            pos += 1
        print("   loads", len(loads), "instructions walked", steps, "first vector-memory wait on the path: vmcnt(%s)" % first_wait, "pairs released by counted waits:", released,
              "instructions touching registers in flight:", len(touched))
        for t in touched[:8]:
            print("      ", t)
        if touched or inflight or first_wait != 61:
            bad += 1
sys.exit(1 if bad else 0)
