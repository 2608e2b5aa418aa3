#!/usr/bin/env python3
"""rocprofv3 --pmc output directory -> mean of every counter per kernel name (dispatches after the first `skip` of a name).
    python3 tools/probes/pmc_by_kernel.py <dir> [skip=4]"""
import collections, csv, glob, os, sys
d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 4
acc = collections.defaultdict(lambda: collections.defaultdict(list))
seen = collections.Counter()
disp = {}
for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        k = int(r["Dispatch_Id"])
        e = disp.setdefault(k, {"name": r["Kernel_Name"].split("(")[0].replace("void ", "").replace("vof::", ""), "c": collections.Counter(),
                                "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "grid": r.get("Grid_Size"), "vgpr": r.get("VGPR_Count"), "sgpr": r.get("SGPR_Count")})
        e["c"][r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(disp):
    e = disp[k]
    seen[e["name"]] += 1
    if seen[e["name"]] <= skip:
        continue
    for cn, cv in e["c"].items():
        acc[e["name"]][cn].append(cv)
    acc[e["name"]]["_us"].append(e["us"])
    acc[e["name"]]["_meta"] = [(e["grid"], e["vgpr"], e["sgpr"])]
for n in sorted(acc, key=lambda n: -sum(acc[n]["_us"])):
    cs = acc[n]
    if len(cs["_us"]) < 2:
        continue
    print("%s: %d dispatches, %.1f us, grid/vgpr/sgpr %s" % (n, len(cs["_us"]), sum(cs["_us"]) / len(cs["_us"]), cs["_meta"][0]))
    print("    " + "  ".join("%s %.4g" % (cn, sum(v) / len(v)) for cn, v in sorted(cs.items()) if not cn.startswith("_")))
