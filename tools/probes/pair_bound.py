#!/usr/bin/env python3
"""What do the two pair kernels of the step (k_tm, k_jacobi_pair) spend their time on?  Diagnostic build
(make -C taichi-2d-vof_amd/csrc wavetimes); every figure is `reps` launches of ONE kernel on the state of a real run,
between one HIP event pair, all in one process on one engine (vof_debug_time_kernel: the state is not advanced).

    python3 tools/probes/pair_bound.py [--n 4096] [--at 96,704] [--reps 10] [--rows] [--map]

 * ablations (wrong values, timing only; bits of ABL_* in kernels/common.h): loads of one fixed row (no HBM stream in),
   no global store, either wave of the pair at s_setprio 1, the second wave idle, the first wave passing its rows on
   without computing;
 * --rows: chunk-length sweeps (knobs tm_rows / jacobi_pair_rows);
 * --map: per-wave start / end stamps and cycles inside barriers of one launch."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "taichi-2d-vof_amd"))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--at", default="96,704")
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("-ic", type=int, default=1)
ap.add_argument("--dt", type=float, default=0.0)
ap.add_argument("--rows", action="store_true")
ap.add_argument("--map", action="store_true")
ap.add_argument("--tm-rows", default="24,32,40,48,56,64,80,96,128")
ap.add_argument("--pair-rows", default="40,56,64,72,80,96,120,160")
ap.add_argument("--no-ablations", action="store_true")
ap.add_argument("--map-real", action="store_true")
ap.add_argument("--param", action="append", default=[])
a = ap.parse_args()
from vof2d import _abi
from vof2d.engine import Engine, make_desc

lib = C.CDLL(os.path.join(ROOT, "taichi-2d-vof_amd", "csrc", "build", "variants", "libvof2d_wavetimes.so"))
api = _abi.bind(lib, "vof_")
tk = lib.vof_debug_time_kernel
tk.restype = C.c_int
tk.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float)]
dbg = lib.vof_debug_wave_times
dbg.restype = C.c_int
dbg.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32]
kw = {"dt": a.dt} if a.dt > 0 else ({"dt": 1e-6} if a.n > 4096 else {})
e = Engine(api, make_desc(api, a.n, a.n, "f64", "f32", device=0, **kw))
e.set_param("fuse_tm", 1)
for kv in a.param:
    e.set_param(kv.split("=")[0], float(kv.split("=")[1]))
e.set_init_F(a.ic)
NAMES = {0: "as shipped", 1: "fixed-row loads", 2: "no store", 3: "fixed-row loads, no store", 4: "wave 0 NOT at prio 1",
         8: "wave 1 at prio 1 too", 16: "wave 1 idle", 32: "wave 0 passes rows on", 48: "wave 0 passes rows on, wave 1 idle",
         19: "wave 1 idle, no HBM stream", 35: "wave 0 passes rows on, no HBM stream",
         64: "momentum wave without the advection / diffusion of u*", 192: "... of u* and v*",
         256: "as shipped, caches flushed before every launch", 257: "fixed-row loads, caches flushed"}
ABLS = (0, 1, 2, 3, 4, 8, 16, 32, 48, 19, 35)
KERNELS = ((0, 1, "k_jacobi_pair (work plan)"), (0, 0, "k_jacobi_pair (uniform chunks)"), (1, 0, "k_tm y first"), (2, 0, "k_tm x first"))


def t(which, abl, plan, reps=None):
    us = C.c_float(0)
    rc = tk(e._h, which, abl, plan, reps or a.reps, C.byref(us))
    assert rc == 0, (rc, which, abl)
    return us.value


def wave_map(which, kid, plan, stride, rows_knob):
    cap = 1 << 15
    assert dbg(e._h, kid, None, cap) == 0
    t(which, 0, plan, 1)
    st = np.zeros((cap, 2), np.uint64)
    cy = np.zeros((cap, 2), np.uint64)
    assert dbg(e._h, kid, st.ctypes.data, cap) == 0
    assert dbg(e._h, -2, cy.ctypes.data, cap) == 0
    assert dbg(e._h, -1, None, cap) == 0          # disarm
    m = st[:, 1] > 0
    ids = np.nonzero(m)[0]
    t0 = st[m, 0].astype(np.int64); t1 = st[m, 1].astype(np.int64)
    base = t0.min()
    dur = (t1 - t0) / 100.0
    span = (t1.max() - base) / 100.0
    wait = cy[m, 0].astype(np.float64); tot = np.maximum(cy[m, 1].astype(np.float64), 1.0)
    role = ids & 1
    print("   %d waves, span %.1f us; wave us: mean %.1f p50 %.1f p90 %.1f max %.1f; last start %.1f us" % (
        len(ids), span, dur.mean(), np.median(dur), np.percentile(dur, 90), dur.max(), (t0.max() - base) / 100.0))
    for r in (0, 1):
        s = role == r
        print("   wave %d of the pairs: share of its cycles inside barriers: mean %.3f p10 %.3f p90 %.3f; cycles per wave mean %.0f (= %.2f GHz x its us)" % (
            r, (wait[s] / tot[s]).mean(), np.percentile(wait[s] / tot[s], 10), np.percentile(wait[s] / tot[s], 90), tot[s].mean(),
            (tot[s] / (dur[s] * 1e3)).mean()))
    edges = np.linspace(0, t1.max() - base, 21)
    mid = (edges[:-1] + edges[1:]) / 2 + base
    print("   waves in flight over the span (20 slices): " + " ".join("%d" % int(((t0 <= x) & (t1 > x)).sum()) for x in mid))
    if plan == 0:
        ntt = (a.n + stride - 1) // stride
        pairs = ids[role == 0] >> 1
        d0 = dur[role == 0]
        nch = int(pairs.max()) // ntt + 1
        D = np.full((nch, ntt), np.nan)
        S = np.full((nch, ntt), np.nan)
        D[pairs // ntt, pairs % ntt] = d0
        S[pairs // ntt, pairs % ntt] = (t0[role == 0] - base) / 100.0
        g = max(1, nch // 16)
        print("   %d chunk rows x %d tile columns; mean pair us per chunk row (groups of %d): %s" % (nch, ntt, g, " ".join("%.0f" % np.nanmean(D[k:k + g]) for k in range(0, nch, g))))
        print("   mean start us per chunk row: %s" % " ".join("%.0f" % np.nanmean(S[k:k + g]) for k in range(0, nch, g)))
        print("   mean pair us per tile column: %s" % " ".join("%.0f" % np.nanmean(D[:, j]) for j in range(ntt)))


done = 0
for at in [int(x) for x in a.at.split(",")]:
    e.step(at - done)
    done = at
    e.sync()
    print("== %d^2 fp64 ic %d after %d steps (tm_steps %d, pair_launches %d)" % (a.n, a.ic, at, e.get_counter("tm_steps"), e.get_counter("pair_launches")), flush=True)
    for which, plan, label in (() if a.no_ablations else KERNELS):
        base = []
        line = []
        for rnd in range(2):
            for abl in ABLS + ((64, 192) if which else ()) + (256, 257):
                us = t(which, abl, plan)
                if abl == 0:
                    base.append(us)
                line.append((rnd, abl, us))
        print(" %s: %s" % (label, " | ".join("%s %.1f / %.1f" % (NAMES[abl], [u for r, b, u in line if b == abl][0], [u for r, b, u in line if b == abl][1])
                                             for abl in ABLS + ((64, 192) if which else ()) + (256, 257))), flush=True)
    if a.map:
        for which, kid, plan, stride, label in ((1, 14, 0, 112, "k_tm y first"), (0, 13, 0, 108, "k_jacobi_pair (uniform chunks)"), (0, 13, 1, 108, "k_jacobi_pair (work plan)")):
            print(" wave map of one launch of %s:" % label)
            wave_map(which, kid, plan, stride, None)
    if a.map_real:    # one launch of k_tm as the batch graphs run it (its work plan included): the last k_tm launch of a 16-step batch
        cap = 1 << 15
        assert dbg(e._h, 14, None, cap) == 0
        e.step(16); done += 16
        e.sync()
        st = np.zeros((cap, 2), np.uint64)
        assert dbg(e._h, 14, st.ctypes.data, cap) == 0
        assert dbg(e._h, -1, None, cap) == 0
        m = st[:, 1] > 0
        t0 = st[m, 0].astype(np.int64); t1 = st[m, 1].astype(np.int64)
        base = t0.min(); dur = (t1 - t0) / 100.0
        edges = np.linspace(0, t1.max() - base, 21); mid = (edges[:-1] + edges[1:]) / 2 + base
        print(" k_tm inside the batch graphs (tm_plan_pairs %d): %d waves, span %.1f us, wave us mean %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f; sum of durations / (span x 3072 slots) = %.2f" % (
            0, len(t0), (t1.max() - base) / 100.0, dur.mean(), np.percentile(dur, 10), np.median(dur), np.percentile(dur, 90), dur.max(),
            dur.sum() / ((t1.max() - base) / 100.0 * 3072)))
        print("   waves in flight (20 slices): " + " ".join("%d" % int(((t0 <= x) & (t1 > x)).sum()) for x in mid), flush=True)
        # what would another dispatch order of the same pairs buy?  Greedy list scheduling of the measured durations on the
        # launch's pair slots (a pair = two consecutive waves), in three orders
        import heapq
        idx = np.nonzero(m)[0]
        dp = np.array([dur[(idx // 2) == q].max() for q in np.unique(idx // 2)])
        slots = int(np.array([int(((t0 <= x) & (t1 > x)).sum()) for x in mid]).max() // 2)

        def span(order):
            h = [0.0] * slots
            heapq.heapify(h)
            end = 0.0
            for q in order:
                s0 = heapq.heappop(h)
                heapq.heappush(h, s0 + dp[q])
                end = max(end, s0 + dp[q])
            return end
        n = len(dp)
        med = np.median(dp)
        two = [q for q in range(n) if dp[q] > 1.2 * med] + [q for q in range(n) if dp[q] <= 1.2 * med]
        for nb in (8, 16, 32):
            edges_b = np.linspace(dp.min(), dp.max() + 1e-9, nb + 1)
            cls = np.minimum(nb - 1, np.searchsorted(edges_b, dp, side="right") - 1)
            order = [q for c in range(nb - 1, -1, -1) for q in range(n) if cls[q] == c]
            print("   ... in %d duration classes, longest class first, dispatch order inside a class: %.1f us" % (nb, span(order)))
        print("   list scheduling of the %d measured pair durations on %d slots: in dispatch order %.1f us, longest first %.1f, the pairs above 1.2 x the median first (%d of them) %.1f; sum / slots = %.1f" % (
            n, slots, span(range(n)), span(np.argsort(-dp)), int((dp > 1.2 * med).sum()), span(two), dp.sum() / slots), flush=True)
    if a.rows:
        for knob, which, vals in (("tm_rows", 1, tuple(int(x) for x in a.tm_rows.split(","))), ("jacobi_pair_rows", 0, tuple(int(x) for x in a.pair_rows.split(",")))):
            res = []
            for v in vals + (0,):
                e.set_param(knob, v)
                res.append("%d: %.1f" % (v, min(t(which, 0, 0), t(which, 0, 0))))
            print(" %s (0 = the heuristic), us per launch, uniform chunks: %s" % (knob, "  ".join(res)), flush=True)
