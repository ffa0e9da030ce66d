#!/usr/bin/env python3
"""profiles/traffic.json from two rocprofv3 --pmc runs of bench.py (FETCH_SIZE pass, WRITE_SIZE pass).

  python tools/traffic_from_pmc.py <fetch_dir> <write_dir> "<PARAMS> k=<parties>" > profiles/traffic.json

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reports half of the
bytes of a wide coalesced read stream and WRITE_SIZE the exact bytes (MI355X_MICROARCH.md, HBM section;
calibrated there for 16-B lanes -- these kernels use 8-B lanes, so treat the read side as approximate).
Counter units are KB.  Kernel names are normalised to the template spelling used by mkhe_prof_name().
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def per_kernel(d, counter):
    out = collections.defaultdict(list)
    for cc in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(cc)):
            if r["Counter_Name"] != counter:
                continue
            m = re.search(r"mkhe::(?:h16::|h32::)?(\w+)(<[^>]*>)?", r["Kernel_Name"])
            if not m:
                continue
            key = m.group(1) + (m.group(2) or "").replace(" ", "")
            if key == "ext_inner_group_kernel" or key.startswith("ext_inner_xy_kernel"):         # same timing class as the per-item form (mkhe_prof_name: "ext_inner_kernel")
                key = "ext_inner_kernel"
            if key.startswith("moddown_merged_kernel"):  # same timing class as the per-product ModDown ("moddown[_batch]_kernel")
                key = "moddown_batch_kernel"
            out[key].append(float(r["Counter_Value"]))
    return out


def main():
    """argv: <fetch_dir> <write_dir> "<workload>" [steps warmup [fetch_dir2 write_dir2]]  -- steps / warmup of the profiled bench.py command are
    recorded so that bench.py can refuse the figures when the launch pattern it measures is not the profiled one.  The second pair of
    directories (the same command with the single-pass forward NTT kernel, MKHE_NTT32=1) only contributes kernels the first pair does not have."""
    fetch, write, workload = sys.argv[1], sys.argv[2], sys.argv[3]
    steps, warmup = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (None, None)
    f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    if len(sys.argv) > 7:
        f2, w2 = per_kernel(sys.argv[6], "FETCH_SIZE"), per_kernel(sys.argv[7], "WRITE_SIZE")
        for k in set(f2) | set(w2):
            if k not in f and k not in w:
                if k in f2: f[k] = f2[k]
                if k in w2: w[k] = w2[k]
    kernels = {}
    for k in sorted(set(f) | set(w)):
        fm = sum(f[k]) / len(f[k]) if f.get(k) else 0.0
        wm = sum(w[k]) / len(w[k]) if w.get(k) else 0.0
        kernels[k] = dict(fetch_size_kb=fm, write_size_kb=wm, launches=len(f.get(k, [])),
                          hbm_bytes_per_launch=(2 * fm + wm) * 1024)
        # the Decompose NTT runs two launch shapes per MulRelin (1792 and 896 limbs) and the engine may settle on a different kernel form for each:
        # the two shapes apart (told by what a dispatch writes; the dispatches of the two passes are in the same order)
        if k in ("ntt16_fwd_kernel<true>", "ntt32_fwd_kernel<true>") and f.get(k) and w.get(k) and len(f[k]) == len(w[k]) and max(w[k]) > 1.5 * min(w[k]):
            thr = 0.5 * (max(w[k]) + min(w[k]))
            for name, sel in (("large", [i for i, v in enumerate(w[k]) if v > thr]), ("small", [i for i, v in enumerate(w[k]) if v <= thr])):
                if sel:
                    fs, ws = sum(f[k][i] for i in sel) / len(sel), sum(w[k][i] for i in sel) / len(sel)
                    kernels[k][name] = dict(fetch_size_kb=fs, write_size_kb=ws, launches=len(sel), hbm_bytes_per_launch=(2 * fs + ws) * 1024)
    # a kernel with a single template instance in the run is also listed under its bare name (bench.py's timing classes
    # do not spell the template arguments of inner_product_kernel<NT>)
    bare = collections.defaultdict(list)
    for k in kernels:
        if "<" in k:
            bare[k.split("<")[0]].append(k)
    for b, ks in bare.items():
        if len(ks) == 1 and b not in kernels:
            kernels[b] = dict(kernels[ks[0]], instance=ks[0])
    import hashlib
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mkhe-kklss_amd", "csrc")
    h = hashlib.sha256()
    for fn in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):      # same digest as bench.py csrc_digest()
        h.update(os.path.basename(fn).encode()); h.update(open(fn, "rb").read())
    json.dump(dict(workload=workload, csrc_sha256=h.hexdigest(), formula="(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch",
                   command="MKHE_NO_OVERLAP=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python3 bench.py --steps %s --warmup %s --no-cpu --no-extras" % (steps, warmup),
                   steps=steps, warmup=warmup, kernels=kernels), sys.stdout, indent=1)


if __name__ == "__main__":
    main()
