#!/usr/bin/env python3
"""Same-PROCESS A/B of a tuning knob: several engines per value live side by side in one process and take turns,
20 profiled steps at a time (vof_profile_steps: one HIP event pair per dispatch), so that the process-level state that
makes k_momentum bimodal between processes (176 / 188 us with the same binary) hits every value alike.

    python3 tools/knob_ab.py knob v0,v1[,...] [--n 4096] [--engines 2] [--rounds 6] [--dtype f64] [-ic 1] [--skip 60]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "taichi-2d-vof_amd"))
ap = argparse.ArgumentParser()
ap.add_argument("knob")
ap.add_argument("values")
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--engines", type=int, default=2)
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--dtype", default="f64")
ap.add_argument("-ic", type=int, default=1)
ap.add_argument("--skip", type=int, default=60)
ap.add_argument("--lib", default="")
ap.add_argument("--dt", type=float, default=0.0)
a = ap.parse_args()
from vof2d import _abi
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc
api = _abi.bind(ctypes.CDLL(a.lib, mode=ctypes.RTLD_GLOBAL), "vof_") if a.lib else hip_api()
vals = [float(v) for v in a.values.split(",")]
engs = []
for k in range(a.engines):
    for v in vals:
        e = Engine(api, make_desc(api, a.n, a.n, a.dtype, "f32", device=0, **({"dt": a.dt} if a.dt > 0 else {})))
        e.set_param(a.knob, v)
        e.set_init_F(a.ic)
        e.step(a.skip)
        engs.append((v, e))
acc = {v: {} for v in vals}
for rnd in range(a.rounds):
    for v, e in engs:
        for k, (us, n) in e.profile_steps(20).items():
            acc[v].setdefault(k, []).append(us)
for v in vals:
    parts = []
    for k in sorted(acc[v]):
        xs = acc[v][k]
        xs = sorted(xs[a.engines:])          # (the first round of every engine: warm-up)
        acc[v][k] = xs
        parts.append("%s %.1f (min %.1f max %.1f)" % (k.replace("k_", ""), xs[len(xs) // 2], xs[0], xs[-1]))
    tot = sum(acc[v][k][len(acc[v][k]) // 2] * (2 if k == "k_jacobi_tb" else 1) for k in acc[v] if k in ("k_momentum", "k_jacobi_tb", "k_transport"))
    print("%s=%g (median of %d x 20 steps): %s | kernels per step %.1f us" % (a.knob, v, (a.rounds - 1) * a.engines, "  ".join(parts), tot))
