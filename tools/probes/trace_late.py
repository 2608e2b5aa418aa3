#!/usr/bin/env python3
"""Kernel durations and the wall span per step from a rocprofv3 --kernel-trace CSV, over the LAST n momentum dispatches.
    python3 tools/probes/trace_late.py <dir or csv> [n=300]"""
import csv, glob, os, sys
src = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
f = src if src.endswith(".csv") else sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("<")[0].replace("void vof::", "")))
rows.sort()
mom = [i for i, r in enumerate(rows) if r[2] == "k_momentum"]
# halves: two k_momentum dispatches per step
per_step = 2 if len(mom) > 1500 else 1
i0 = mom[-n * per_step]
sel = rows[i0:]
acc = {}
for s, e, k in sel:
    a = acc.setdefault(k, [0, 0]); a[0] += e - s; a[1] += 1
span = (max(e for s, e, k in sel) - sel[0][0]) / n
busy = 0; cur_e = sel[0][0]
for s, e, k in sel:
    if e > cur_e: busy += e - max(s, cur_e); cur_e = e
print("%s: last %d steps: span %.1f us/step, covered by at least one kernel %.1f us/step | " % (os.path.basename(os.path.dirname(f)) or f, n, span / 1e3, busy / n / 1e3)
      + "  ".join("%s %.1f us x %.1f" % (k, a[0] / a[1] / 1e3, a[1] / n) for k, a in sorted(acc.items())))
