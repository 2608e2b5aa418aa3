#!/usr/bin/env python3
"""rocprofv3 --pmc passes of tools/collect_bound.sh -> per-kernel, per-window counter averages.

    python3 tools/summarize_bound.py reduce <pass dir> <out.json>      (on the GPU box: CSV -> small JSON)
    python3 tools/summarize_bound.py table <tag> [--nx 4096 --ny 4096 --esz 8]   (here: JSONs -> profiles/<tag>_bound.md)

A dispatch belongs to step n when n k_momentum dispatches have started up to and including it (tools/bound_run.py
launches nothing but vof_step).  Windows: steps 11-60 (before the tiny-value front) and 301-350 (inside it)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WINDOWS = {"11-60": (11, 60), "301-350": (301, 350)}


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("vof::", "").strip()
    if n.startswith("k_transport"):
        return "k_transport<yfirst>" if "true" in n else "k_transport<xfirst>"
    return n.split("<")[0]


def reduce_pass(d, out):
    paths = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not paths:
        json.dump({"error": "no counter_collection.csv under " + d}, open(out, "w"))
        return
    disp = {}
    for r in csv.DictReader(open(paths[0])):
        k = int(r["Dispatch_Id"])
        e = disp.setdefault(k, {"name": short(r["Kernel_Name"]), "t0": int(r["Start_Timestamp"]), "t1": int(r["End_Timestamp"]),
                                "vgpr": r.get("VGPR_Count"), "sgpr": r.get("SGPR_Count"), "grid": r.get("Grid_Size"), "c": {}})
        e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    step = 0
    acc = {w: collections.defaultdict(lambda: collections.defaultdict(list)) for w in WINDOWS}
    meta = {}
    for k in sorted(disp, key=lambda i: disp[i]["t0"]):
        e = disp[k]
        if e["name"] == "k_momentum":
            step += 1
        for w, (lo, hi) in WINDOWS.items():
            if lo <= step <= hi and e["name"].startswith("k_"):
                for cn, cv in e["c"].items():
                    acc[w][e["name"]][cn].append(cv)
                acc[w][e["name"]]["_us"].append((e["t1"] - e["t0"]) / 1e3)
                meta[e["name"]] = {"vgpr": e["vgpr"], "sgpr": e["sgpr"], "grid": e["grid"]}
    res = {"steps_seen": step, "meta": meta, "windows": {}}
    for w in acc:
        res["windows"][w] = {k: {cn: sum(v) / len(v) for cn, v in cs.items()} | {"_n": len(cs["_us"])} for k, cs in acc[w].items()}
    json.dump(res, open(out, "w"), indent=1)
    print(out, "steps", step, {k: v["_n"] for k, v in res["windows"]["11-60"].items()})


def table(tag, nx, ny, esz):
    merged = {w: collections.defaultdict(dict) for w in WINDOWS}
    us = {w: collections.defaultdict(list) for w in WINDOWS}
    meta = {}
    for p in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag + "_bound_*.json"))):
        r = json.load(open(p))
        if "error" in r:
            continue
        meta.update(r.get("meta", {}))
        for w in r["windows"]:
            for k, cs in r["windows"][w].items():
                for cn, cv in cs.items():
                    if cn == "_us":
                        us[w][k].append(cv)
                    elif cn != "_n":
                        merged[w][k][cn] = cv
    cells = nx * ny
    out = os.path.join(ROOT, "profiles", tag + "_bound.json")
    json.dump({"meta": meta, "windows": {w: {k: dict(v, _us_by_pass=us[w][k]) for k, v in merged[w].items()} for w in merged},
               "nx": nx, "ny": ny}, open(out, "w"), indent=1)
    print("wrote", out)
    for w in merged:
        print("== window", w)
        for k, c in merged[w].items():
            g = c.get
            print(k, "us (per pass):", " ".join("%.1f" % x for x in us[w][k]))
            for cn in sorted(c):
                print("   %-36s %.4g   per cell %.3f" % (cn, c[cn], c[cn] / cells))


if __name__ == "__main__":
    if sys.argv[1] == "reduce":
        reduce_pass(sys.argv[2], sys.argv[3])
    else:
        import argparse
        ap = argparse.ArgumentParser()
        ap.add_argument("cmd")
        ap.add_argument("tag")
        ap.add_argument("--nx", type=int, default=4096)
        ap.add_argument("--ny", type=int, default=4096)
        ap.add_argument("--esz", type=int, default=8)
        a = ap.parse_args()
        table(a.tag, a.nx, a.ny, a.esz)
