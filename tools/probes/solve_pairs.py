#!/usr/bin/env python3
"""The residual-terminated pressure solve and back-to-back sweeps with ten sweeps per launch (k_jacobi_pair, knob solve_pairs)
against five (k_jacobi_tb): us per sweep, and the solve of BASELINE configs[1] (1024^2 to a relative residual of 1e-6)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "taichi-2d-vof_amd"))
import numpy as np
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc
api = hip_api()
for n in (1024, 2048, 4096):
    res = {}
    for sp in (0, 1):
        e = Engine(api, make_desc(api, n, n, "f64", "f32", device=0))
        e.set_param("solve_pairs", sp)
        e.set_init_F(1)
        e.step(1)
        us = min(e.time_jacobi(200) for _ in range(3)) * 1e3
        line = "%d^2 solve_pairs %d: %.2f us per sweep back to back" % (n, sp, us)
        if n == 1024:
            e2 = Engine(api, make_desc(api, n, n, "f64", "f32", device=0))
            e2.set_param("solve_pairs", sp)
            e2.set_init_F(1)
            e2.cal_nu_rho(); e2.get_normal_young(); e2.advect_upwind(); e2.set_BC()
            e2.solve_p_jacobi(10)
            e2.sync(); t0 = time.perf_counter()
            it, r = e2.solve_p(1e-6, 3000000, 5000, "rel")
            e2.sync(); dt = time.perf_counter() - t0
            line += "; solve to 1e-6 (rel): %d sweeps, residual %.3e, %.3f s" % (it, r, dt)
            res[sp] = e2.get("p")
        print(line, flush=True)
    if len(res) == 2:
        print("   p of the two solves equal: %s" % bool(np.array_equal(res[0], res[1])))
