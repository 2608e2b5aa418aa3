#!/usr/bin/env python3
"""Per-step wall span against the kernels' own durations, from rocprofv3 --kernel-trace CSVs of tools/bound_run.py:
what the two-chain batch graphs buy, seen from the GPU's own timestamps.

    python3 tools/summarize_overlap_trace.py <tag> <label>=<dir or kernel_trace.csv> [<label>=<...> ...] [--windows 21-220,301-600,701-1000]

Writes profiles/<tag>_overlap_trace.md.  A step is found by its k_momentum dispatches (one per step in the one-chain
schedule, two in the two-chain one); the span of a window runs from the first dispatch of its first step to the end of
the last dispatch of its last step."""
import argparse, csv, glob, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("tag")
ap.add_argument("traces", nargs="+")
ap.add_argument("--windows", default="21-220,301-600,701-1000")
ap.add_argument("--cmd", default="")
ap.add_argument("--steps", type=int, default=1000, help="steps of the traced run")
a = ap.parse_args()
wins = [tuple(int(x) for x in w.split("-")) for w in a.windows.split(",")]
lines = ["# Wall span per step against the kernels' own durations (%s)" % a.tag, ""]
if a.cmd:
    lines += ["command: `%s`" % a.cmd, ""]
lines += ["Timestamps of `rocprofv3 --kernel-trace`; `covered` = time during which at least one kernel runs; `kernel time` = sum of",
          "the dispatches' own durations (above the span where kernels overlap).", "",
          "| run | steps | span us/step | covered us/step | kernel time us/step | dispatches: avg us x per step |", "|---|---|---|---|---|---|"]
for spec in a.traces:
    label, src = spec.split("=", 1)
    f = src if src.endswith(".csv") else sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("<")[0].replace("void vof::", "").strip()))
    rows.sort()
    mom = [i for i, r in enumerate(rows) if r[2] in ("k_momentum", "k_tm")]     # (one dispatch per step starts with either)
    per_step = 2 if len(mom) > 1.5 * a.steps else 1      # (two-chain: two dispatches per kernel and step; the first step
    for st0, st1 in wins:                                #  and what no batch holds run one-chain: a window's ends are off by a step at most)
        n = st1 - st0 + 1
        i0 = mom[min((st0 - 1) * per_step, len(mom) - 1)]
        i1 = mom[st1 * per_step] if st1 * per_step < len(mom) else len(rows)
        sel = rows[i0:i1]
        acc = {}
        for s, e, k in sel:
            x = acc.setdefault(k, [0, 0]); x[0] += e - s; x[1] += 1
        span = (max(e for s, e, k in sel) - sel[0][0]) / n
        busy, cur = 0, sel[0][0]
        for s, e, k in sel:
            if e > cur:
                busy += e - max(s, cur); cur = e
        ksum = sum(x[0] for x in acc.values()) / n
        lines.append("| %s | %d-%d | %.1f | %.1f | %.1f | %s |" % (label, st0, st1, span / 1e3, busy / n / 1e3, ksum / 1e3,
                     ", ".join("%s %.1f x %.1f" % (k, x[0] / x[1] / 1e3, x[1] / n) for k, x in sorted(acc.items()))))
out = os.path.join(ROOT, "profiles", a.tag + "_overlap_trace.md")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
