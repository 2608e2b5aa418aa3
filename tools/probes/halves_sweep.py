#!/usr/bin/env python3
"""Wall ms/step (graph replay) of several knob settings side by side in one process: E engines per setting take turns,
medians over the rounds (an engine's k_momentum has its own speed, profiles/r04_bound.md section 6).
    python3 tools/probes/halves_sweep.py "overlap_halves=0" "overlap_halves=1" "overlap_halves=1,batch_steps=16" ...
      [--n 4096] [--dtype f64] [-ic 1] [--skip 60] [--steps 80] [--rounds 5] [--engines 2]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "taichi-2d-vof_amd"))
ap = argparse.ArgumentParser()
ap.add_argument("settings", nargs="+")
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--ny", type=int, default=0)
ap.add_argument("--dtype", default="f64")
ap.add_argument("-ic", type=int, default=1)
ap.add_argument("--skip", type=int, default=60)
ap.add_argument("--steps", type=int, default=80)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--engines", type=int, default=2)
ap.add_argument("--dt", type=float, default=0.0)
ap.add_argument("--lib", default="", help="a variant build of the library (csrc/build/variants/...)")
a = ap.parse_args()
import ctypes
from vof2d import _abi
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc
api = _abi.bind(ctypes.CDLL(a.lib, mode=ctypes.RTLD_GLOBAL), "vof_") if a.lib else hip_api()
engs = []
for k in range(a.engines):
    for st in a.settings:
        e = Engine(api, make_desc(api, a.n, a.ny or a.n, a.dtype, "f32", device=0, **({"dt": a.dt} if a.dt > 0 else {})))
        for kv in st.split(","):
            if kv:
                name, v = kv.split("=")
                e.set_param(name, float(v))
        e.set_init_F(a.ic)
        e.step(a.skip); e.sync()
        engs.append((st, e))
acc = {st: [] for st in a.settings}
for r in range(a.rounds):
    for st, e in engs:
        e.sync(); t0 = time.perf_counter(); e.step(a.steps); e.sync()
        acc[st].append(1e3 * (time.perf_counter() - t0) / a.steps)
for st in a.settings:
    xs = sorted(acc[st])
    print("%-60s median %.4f  min %.4f  max %.4f ms/step" % (st or "(default)", xs[len(xs) // 2], xs[0], xs[-1]), flush=True)
