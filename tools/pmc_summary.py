"""Summarise rocprofv3 --pmc CSVs per kernel: mean counter value per dispatch + duration."""
import csv, sys, glob, collections, os
for d in sys.argv[1:]:
    for cc in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        kt = glob.glob(os.path.join(os.path.dirname(cc), "*kernel_trace.csv"))
        dur = {}
        if kt:
            for r in csv.DictReader(open(kt[0])):
                dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        meta = {}
        for r in csv.DictReader(open(cc)):
            k = r["Kernel_Name"].split("(")[0][-40:]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] in dur:
                agg[k]["_dur_ns"].append(dur[r["Dispatch_Id"]])
            meta[k] = (r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["Scratch_Size"], r["LDS_Block_Size"])
        for k, c in agg.items():
            if "ntt" not in k and "inner" not in k and "moddown" not in k: continue
            n = len(next(iter(c.values())))
            print(k, "grid/wg/vgpr/agpr/scratch/lds", meta[k], "dispatches", n)
            for name, v in sorted(c.items()):
                # take dispatches after warm-up: last third
                vv = v[len(v) * 2 // 3:]
                print("    %-24s %.4g" % (name, sum(vv) / len(vv)))
