#!/usr/bin/env python3
"""GPU tuning sweep of the Jacobi kernels: sweeps fused per launch (tb) x rows per wave chunk.
Checks that every variant reproduces the single-sweep kernel's p exactly, then times it."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "taichi-2d-vof_amd"))
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dtype = sys.argv[2] if len(sys.argv) > 2 else "f64"
api = hip_api()
e = Engine(api, make_desc(api, n, n, dtype, "f32"))
e.set_init_F(1)
e.step(3)                       # realistic p / rhs
for v in ("cal_nu_rho", "get_normal_young", "advect_upwind", "set_BC"):
    getattr(e, v)()
p0 = e.get("p")
e.set_param("jacobi_tb", 1)
e.solve_p_jacobi(20)
ref = e.get("p")
esz = 8 if dtype == "f64" else 4
for tb, rows_list in ((1, (0,)), (2, (16, 32, 64)), (5, (16, 32, 64, 128)), (10, (32, 64, 128, 256))):
    for rows in rows_list:
        e.set("p", p0)
        e.set_param("jacobi_tb", tb)
        e.set_param("jacobi_tb_rows", rows)
        e.solve_p_jacobi(20)
        ok = np.array_equal(e.get("p"), ref)
        ms = e.time_jacobi(200)
        print("tb=%2d rows=%3d  same_as_single=%s  %.2f us/sweep  %.0f GB/s-equivalent (24B rule/sweep)  launch=%.1f us" % (
            tb, rows, ok, ms * 1e3, 3 * esz * n * n / (ms * 1e-3) / 1e9, ms * 1e3 * tb), flush=True)
