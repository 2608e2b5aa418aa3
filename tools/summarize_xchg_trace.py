#!/usr/bin/env python3
"""Kernel timeline of captured strip steps with the in-library RCCL exchange (loopback on one GPU):
rocprofv3 --kernel-trace CSV of `VOF_COMM_LOOPBACK=1 ... tools/p2p_overhead.py --modes compute,native-fused`
-> profiles/<tag>_strip_exchange_timeline.md: three consecutive steps, every kernel with queue, start, end, and
how much of the send/recv group's kernel runs under the transport of the inner rows.

    python3 tools/summarize_xchg_trace.py <kernel_trace.csv> <tag>"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(n):
    return n.replace("void ", "").replace("vof::", "").split("(")[0][:48]


def main_pairs(rows, tag):
    """Overlap mode 5: a middle step is k_jacobi_pair, then k_tm on the edge bands and the send / recv kernel on the
    communication stream beside k_tm on the other rows."""
    ncc = [i for i, r in enumerate(rows) if "nccl" in r["Kernel_Name"].lower()]
    i = ncc[(3 * len(ncc)) // 4]       # (inside the second of two rounds: the middle of the list is where a call ends)
    while "k_jacobi_pair" not in rows[i]["Kernel_Name"]:
        i -= 1
    sel = rows[i:i + 4 * 14]
    t0 = int(sel[0]["Start_Timestamp"])
    T = lambda r: ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3)
    out = ["# Strip step with the in-library RCCL exchange (mode 5: the pair kernels, two middle steps per captured graph), neighbours looped back (%s)\n" % tag,
           "`rocprofv3 --kernel-trace -- python3 tools/p2p_overhead.py --steps 100 --modes compute-pairs,native-pairs --rounds 2`: the interior strip of 8 of an 8192^2 fp64 dam-break (1024 owned rows + 2 x 18 halo rows), three consecutive middle steps of the native-pairs run:\n",
           "| start us | end us | queue | kernel |", "|---|---|---|---|"]
    pairs_i = [k for k, r in enumerate(sel) if "k_jacobi_pair" in r["Kernel_Name"]]
    for k, r in enumerate(sel[:pairs_i[3] if len(pairs_i) > 3 else len(sel)]):
        s, e = T(r)
        out.append("| %.1f | %.1f | %s | %s |" % (s, e, r["Queue_Id"], short(r["Kernel_Name"])))
    dur, hidden, exposed, per, band, rest, jp = [], [], [], [], [], [], []
    for a, b in zip(pairs_i[:-1], pairs_i[1:]):
        step = sel[a:b]
        n = [r for r in step if "nccl" in r["Kernel_Name"].lower()]
        tm = sorted((r for r in step if "k_tm" in r["Kernel_Name"]), key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        if len(n) != 1 or len(tm) < 2:
            continue
        (s, e), (rs, re) = T(n[0]), T(tm[-1])          # (the longest k_tm launch: the rows between the bands)
        dur.append(e - s); hidden.append(max(0.0, min(e, re) - max(s, rs))); exposed.append(max(0.0, e - re))
        band.append(sum(T(x)[1] - T(x)[0] for x in tm[:-1])); rest.append(re - rs); jp.append(T(step[0])[1] - T(step[0])[0])
        per.append(T(sel[b])[0] - T(sel[a])[0])
    m = lambda x: sum(x) / len(x)
    out.append("\nOver %d middle steps: `k_jacobi_pair` (all stored rows) %.1f us, `k_tm` on the two edge bands %.1f us, the send / recv kernel %.1f us -- %.1f us of it under `k_tm` of the other rows (%.1f us) on the other queue, %.1f us after that kernel has ended; step period %.1f us." % (
        len(dur), m(jp), m(band), m(dur), m(hidden), m(rest), m(exposed), m(per)))
    path = os.path.join(ROOT, "profiles", tag + "_strip_exchange_timeline.md")
    open(path, "w").write("\n".join(out) + "\n")
    print("\n".join(out))


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if len(sys.argv) > 3 and sys.argv[3] == "--pairs":
        return main_pairs(rows, sys.argv[2])
    ncc = [i for i, r in enumerate(rows) if "nccl" in r["Kernel_Name"].lower()]
    i = ncc[len(ncc) // 2]
    while "k_momentum" not in rows[i]["Kernel_Name"]:
        i -= 1
    sel = rows[i:i + 6 * 12]
    t0 = int(sel[0]["Start_Timestamp"])
    out = ["# Strip step with the in-library RCCL exchange (mode 4, two steps per captured graph), neighbours looped back (%s)\n" % sys.argv[2],
           "`VOF_COMM_LOOPBACK=1 rocprofv3 --kernel-trace --memory-copy-trace -- python3 tools/p2p_overhead.py --steps 60 --modes compute,native-fused --rounds 1`: the interior strip of 8 of an 8192^2 fp64 dam-break (1024 + 2 x 16 rows), the real byte counts and RCCL operations, device-local copies instead of xGMI.\n",
           "| start us | end us | queue | kernel |", "|---|---|---|---|"]
    steps, hidden, exposed, dur = 0, [], [], []
    for k, r in enumerate(sel):
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        if steps < 3:
            out.append("| %.1f | %.1f | %s | %s |" % (s, e, r["Queue_Id"], short(r["Kernel_Name"])))
        if "nccl" in r["Kernel_Name"].lower() and k + 1 < len(sel) and "k_transport" in sel[k + 1]["Kernel_Name"]:
            ts, te = (int(sel[k + 1]["Start_Timestamp"]) - t0) / 1e3, (int(sel[k + 1]["End_Timestamp"]) - t0) / 1e3
            hidden.append(max(0.0, min(e, te) - max(s, ts)))
            exposed.append(max(0.0, e - te))
            dur.append(e - s)
            steps += 1
    mom = [r for r in sel if "k_momentum" in r["Kernel_Name"]]
    per = [(int(b["Start_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3 for a, b in zip(mom[:-1], mom[1:])]
    per = [x for x in per if x < 2 * min(per)]
    out.append("\nOver the %d steps that follow: the send/recv kernel runs %.1f us per step, %.1f us of it under `k_transport` of the inner rows on the other queue, %.1f us after that kernel has ended; step period (k_momentum to k_momentum) %.1f us." % (
        len(hidden), sum(dur) / len(dur), sum(hidden) / len(hidden), sum(exposed) / len(exposed), sum(per) / len(per)))
    path = os.path.join(ROOT, "profiles", sys.argv[2] + "_strip_exchange_timeline.md")
    open(path, "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
