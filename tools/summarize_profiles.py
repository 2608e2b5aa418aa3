#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (gpurun_out/...) into the small summaries committed under profiles/.

    python tools/summarize_profiles.py <tag> <kernel_stats.csv> [<pmc FETCH_SIZE csv> <pmc WRITE_SIZE csv>] \
        [--nx 4096 --ny 4096 --dtype f64]

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are
collected in separate passes (TCC slots), are in KiB, and on gfx950 FETCH_SIZE counts a wide
(16 B/lane) coalesced streaming read at exactly half its bytes -> hbm = 2*FETCH + WRITE.
"""
import argparse
import collections
import csv
import json
import re
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("stats")
    ap.add_argument("fetch", nargs="?")
    ap.add_argument("write", nargs="?")
    ap.add_argument("--nx", type=int, default=4096)
    ap.add_argument("--ny", type=int, default=4096)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--cmd", default="")
    ap.add_argument("--no-json", action="store_true", help="do not rewrite profiles/jacobi_pmc.json (a pass over another schedule)")
    ap.add_argument("--tm-json", action="store_true", help="the passes are of the k_tm form: write profiles/tm_pmc.json (k_tm, k_jacobi_pair) instead")
    a = ap.parse_args()
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    rows = list(csv.DictReader(open(a.stats)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(os.path.join(out, a.tag + "_kernel_stats.md"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (%s)\n\n" % a.tag)
        if a.cmd:
            f.write("command: `%s`\n\n" % a.cmd)
        f.write("| kernel | calls | avg us | min us | max us | total ms | % |\n|---|---|---|---|---|---|---|\n")
        for r in rows:
            f.write("| %s | %s | %.2f | %.2f | %.2f | %.3f | %.1f |\n" % (
                short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
                float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
    if a.fetch and a.write:
        agg = {}
        for key, path in (("fetch_kib", a.fetch), ("write_kib", a.write)):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(path)):
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
            for k, v in acc.items():
                agg.setdefault(k, {})[key] = sum(v) / len(v)
                agg[k]["launches_" + key] = len(v)
        esz = 8 if a.dtype == "f64" else 4
        with open(os.path.join(out, a.tag + "_hbm_pmc.md"), "w") as f:
            f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), %dx%d %s (%s)\n\n" % (
                a.nx, a.ny, a.dtype, a.tag))
            f.write("hbm bytes/launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024  (gfx950 FETCH_SIZE half-count "
                    "correction for 16 B/lane streaming reads, MI355X_MICROARCH.md HBM section)\n\n")
            f.write("| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM MB/launch | in array passes (%d x %d x %d B) |\n"
                    "|---|---|---|---|---|\n" % (a.nx, a.ny, esz))
            for k, v in agg.items():
                if "fetch_kib" in v and "write_kib" in v:
                    hbm = (2 * v["fetch_kib"] + v["write_kib"]) * 1024
                    f.write("| %s | %.0f | %.0f | %.1f | %.2f |\n" % (k, v["fetch_kib"], v["write_kib"], hbm / 1e6,
                                                                     hbm / (a.nx * a.ny * esz)))
        if a.tm_json:
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "taichi-2d-vof_amd"))
            from vof2d._lib import kernel_source_hash
            rec = {"nx": a.nx, "ny": a.ny, "dtype": a.dtype, "tag": a.tag, "hbm_bytes_per_launch": {}, "kernel_source_sha256": kernel_source_hash(),
                   "rule": "(2*FETCH_SIZE + WRITE_SIZE)*1024, separate --pmc passes", "kernels": {}}
            # (k_tm's fourth template argument: the launch also stores u and v -- the last of a batch, "k_tm_uv" in the in-situ profiles)
            for key, pat in (("tm", r"k_tm<[^,]+, \d+, (true|false), false,"), ("tm_uv", r"k_tm<[^,]+, \d+, (true|false), true,"), ("pair", r"k_jacobi_pair<"),
                             ("momentum", r"k_momentum<"), ("transport", r"k_transport<"), ("tb", r"k_jacobi_tb<")):
                vs = [(2 * v["fetch_kib"] + v["write_kib"]) * 1024 for k, v in agg.items() if re.search(pat, k) and "fetch_kib" in v and "write_kib" in v]
                if vs:
                    rec["hbm_bytes_per_launch"][key] = sum(vs) / len(vs)
                    rec["kernels"][key] = [k for k in agg if re.search(pat, k)]
            json.dump(rec, open(os.path.join(out, "tm_pmc.json"), "w"), indent=1)
            return
        if a.no_json:
            return
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "taichi-2d-vof_amd"))
        from vof2d._lib import kernel_source_hash
        rec = {"nx": a.nx, "ny": a.ny, "dtype": a.dtype, "tag": a.tag, "hbm_bytes_per_launch": {},
               "kernel_source_sha256": kernel_source_hash(),   # bench.py quotes these numbers only for these sources
               "rule": "(2*FETCH_SIZE + WRITE_SIZE)*1024, separate --pmc passes", "kernels": {}}
        for k, v in agg.items():
            if "fetch_kib" in v and "write_kib" in v and "k_jacobi" in k:
                key = "tb" if "k_jacobi_tb" in k else "single"
                rec["hbm_bytes_per_launch"][key] = (2 * v["fetch_kib"] + v["write_kib"]) * 1024
                rec["kernels"][key] = {"name": k, "fetch_size_kib": v["fetch_kib"], "write_size_kib": v["write_kib"]}
        for key, pat in (("momentum", "k_momentum<"), ("transport", "k_transport<")):   # (the other kernels of the four-kernel step: bench.py's step traffic)
            vs = [(2 * v["fetch_kib"] + v["write_kib"]) * 1024 for k, v in agg.items() if pat in k and "fetch_kib" in v and "write_kib" in v]
            if vs:
                rec["hbm_bytes_per_launch"][key] = sum(vs) / len(vs)
                rec["kernels"][key] = [k for k in agg if pat in k]
        json.dump(rec, open(os.path.join(out, "jacobi_pmc.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
