#!/usr/bin/env python3
"""Issue counters per kernel from rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
passes (tools/collect_profiles.sh), side by side for two precisions of one workload.

    python tools/summarize_sq.py <tag> <label=counter_collection.csv> ... --nx 2048 --ny 2048

Per launch: wave instructions (VALU, SALU), per cell of the grid and -- for the fused Jacobi kernel,
which applies five sweeps per launch -- per cell and sweep; share of wave cycles spent waiting."""
import argparse
import collections
import csv
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SWEEPS = {"k_jacobi_tb": 5}


def short(name):
    return name.split("(")[0].replace("void ", "").replace("vof::", "").strip()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("runs", nargs="+", help="label=path")
    ap.add_argument("--nx", type=int, default=2048)
    ap.add_argument("--ny", type=int, default=2048)
    ap.add_argument("--cmd", default="")
    a = ap.parse_args()
    cells = a.nx * a.ny
    out = os.path.join(ROOT, "profiles", a.tag + "_sq_counters.md")
    with open(out, "w") as f:
        f.write("# rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES (%s), %dx%d\n\n" % (a.tag, a.nx, a.ny))
        if a.cmd:
            f.write("command: `%s`\n\n" % a.cmd)
        f.write("Counters are per dispatch, averaged over the launches of the run; instructions are wave instructions "
                "(one per 64 lanes), so `per cell` = instructions x 64 / (V = 2 columns per lane ...) is NOT applied: "
                "`VALU / cell` below is wave instructions x 64 lanes / cells of the grid, i.e. lane-instructions per cell, "
                "redundant lead-in rows and overlap columns included.\n\n")
        f.write("| run | kernel | launches | avg us | VALU / cell | SALU (wave) / cell x 64 | VALU / cell / sweep | waiting share of wave cycles |\n|---|---|---|---|---|---|---|---|\n")
        for run in a.runs:
            label, path = run.split("=", 1)
            acc = collections.defaultdict(lambda: collections.defaultdict(list))
            dur = collections.defaultdict(list)
            seen = set()
            for r in csv.DictReader(open(path)):
                k = short(r["Kernel_Name"])
                if not k.startswith("k_"):
                    continue
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if r["Dispatch_Id"] not in seen:
                    seen.add(r["Dispatch_Id"])
                    dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            for k in sorted(acc, key=lambda n: -sum(dur[n])):
                c = {n: sum(v) / len(v) for n, v in acc[k].items()}
                if len(dur[k]) < 5:
                    continue
                base = k.split("<")[0]
                valu = c.get("SQ_INSTS_VALU", 0) * 64 / cells
                salu = c.get("SQ_INSTS_SALU", 0) * 64 / cells
                wait = c.get("SQ_WAIT_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1)
                f.write("| %s | %s | %d | %.1f | %.1f | %.1f | %s | %.2f |\n" % (
                    label, k, len(dur[k]), sum(dur[k]) / len(dur[k]), valu, salu,
                    ("%.1f" % (valu / SWEEPS[base])) if base in SWEEPS else "", wait))
    print(open(out).read())


if __name__ == "__main__":
    main()
