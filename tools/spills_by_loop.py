#!/usr/bin/env python3
"""Where do a kernel's SGPR spills sit?  Reads the ISA `make -C taichi-2d-vof_amd/csrc asm` leaves in build/asm/*.s and counts, per
loop of each selected kernel (the compiler annotates every block with the header and depth of the innermost loop around it), the
instructions and the v_readlane / v_writelane among them (SGPR spills live in lanes of a VGPR: a v_writelane spills, a v_readlane
restores; the cross-lane reads the kernels do on purpose are DPP moves or v_readfirstlane, not v_readlane).

    python3 tools/spills_by_loop.py [asm.s] [substring of the mangled kernel name ...]        -> markdown table on stdout"""
import collections
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
path = args.pop(0) if args and args[0].endswith(".s") else sorted(glob.glob(os.path.join(ROOT, "taichi-2d-vof_amd", "csrc", "build", "asm", "*gfx950.s")))[0]
pats = args or ["k_tmIdLi2ELb1ELb0ELb1ELi0", "k_tmIdLi2ELb0ELb0ELb1ELi0", "k_jacobi_pairIdLi2ELi5ELb1ELi0", "k_momentumIdLi2ELb1", "k_transportIdLi2ELb1", "k_jacobi_tbIdLi2ELi5ELb1ELb0ELb1"]


def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void vof::", "")
    except Exception:
        return n


lines = open(path).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_ZN3vof\w+:", l)]
print("| kernel | loop (first .. last line of its body in the kernel's listing) | instructions | of them scalar | v_readlane | v_writelane | s_barrier | memory / LDS |")
print("|---|---|---|---|---|---|---|---|")
for k, (i0, name) in enumerate(starts):
    if not any(p in name for p in pats):
        continue
    i1 = next(j for j in range(i0, len(lines)) if lines[j].strip().startswith("s_endpgm"))
    body = lines[i0:i1 + 1]
    labels = {}
    for n, l in enumerate(body):
        t = l.strip()
        if t.startswith(".LBB") and ":" in t:
            labels[t.split(":")[0]] = n
    # loops = ranges closed by a backward branch (the asm printer's own loop comments miss the irreducible ones)
    loops = []
    for n, l in enumerate(body):
        t = l.strip().split()
        if len(t) == 2 and t[0].startswith(("s_cbranch", "s_branch")) and t[1] in labels and labels[t[1]] <= n:
            loops.append((labels[t[1]], n))
    merged = []
    for a, b in sorted(loops):
        if merged and a == merged[-1][0]:
            merged[-1] = (a, max(b, merged[-1][1]))
        else:
            merged.append((a, b))
    loops = merged

    def innermost(n):
        best = None
        for a, b in loops:
            if a <= n <= b and (best is None or (b - a) < (best[1] - best[0])):
                best = (a, b)
        return best
    stats = collections.OrderedDict()
    for n, l in enumerate(body):
        t = l.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        s_ = stats.setdefault(innermost(n), collections.Counter())
        s_["n"] += 1
        s_["scalar"] += op.startswith("s_")
        s_["rl"] += op.startswith("v_readlane")
        s_["wl"] += op.startswith("v_writelane")
        s_["bar"] += op == "s_barrier"
        s_["mem"] += op.startswith(("buffer_", "global_", "ds_", "flat_", "scratch_"))
    short = demangle(name)
    tot = collections.Counter()
    for key, s_ in sorted(stats.items(), key=lambda kv: (-1, -1) if kv[0] is None else kv[0]):
        tot.update(s_)
        if s_["n"] >= 60 or s_["rl"] or s_["wl"]:
            where = "(outside every loop)" if key is None else "lines %d .. %d" % key
            print("| `%s` | %s | %d | %d | %d | %d | %d | %d |" % (short, where, s_["n"], s_["scalar"], s_["rl"], s_["wl"], s_["bar"], s_["mem"]))
    print("| `%s` | **whole kernel** | %d | %d | %d | %d | %d | %d |" % (short, tot["n"], tot["scalar"], tot["rl"], tot["wl"], tot["bar"], tot["mem"]))
