#!/usr/bin/env python3
"""Cost of the tiny-numerator tier of div_by_const in the fused Jacobi: time n sweeps with p filled
with ordinary values, with 1e-300 (scaled tier), with 1e-320 (subnormal quotients), with zeros."""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "taichi-2d-vof_amd"))
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc
api = hip_api()
n = 4096
e = Engine(api, make_desc(api, n, n, "f64", "f32", device=0)); e.set_init_F(1); e.step(2)
rng = np.random.default_rng(0)
base = rng.uniform(1.0, 2.0, (n + 2, n + 2))
for name, scale in (("ordinary", 1.0), ("1e-295", 1e-295), ("1e-310", 1e-310), ("zero", 0.0), ("ordinary", 1.0)):
    e.set("p", base * scale)
    for f in ("u", "v"):
        e.set(f, np.zeros((n + 2, n + 2)))
    e.advect_upwind()          # u*, v* = gravity only -> rhs ordinary in liquid ... keep rhs tiny too:
    e.set("u_star", np.zeros((n + 2, n + 2))); e.set("v_star", np.zeros((n + 2, n + 2)))
    e.jacobi_sweeps_residual(2, build_rhs=True)   # builds rhs = 0
    e.set("p", base * scale)
    ts = [1e3 * e.time_jacobi(10) * 5 for _ in range(3)]
    print("%-9s fused launch (5 sweeps): %s us" % (name, " ".join("%.1f" % t for t in ts)), flush=True)
