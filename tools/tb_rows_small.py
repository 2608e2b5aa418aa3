#!/usr/bin/env python3
"""Fused-Jacobi chunk length on small (cache-resident) grids: us per sweep of k_jacobi_tb back to back.
    python3 tools/tb_rows_small.py [n=1024] [rows=0,4,6,8,12,16,24,32]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "taichi-2d-vof_amd"))
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rows = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,4,6,8,12,16,24,32").split(",")]
api = hip_api()
for r in rows + rows[:1]:
    e = Engine(api, make_desc(api, n, n, "f64", "f32", device=0))
    e.set_init_F(1)
    e.cal_nu_rho(); e.get_normal_young(); e.advect_upwind(); e.set_BC()
    e.set_param("jacobi_tb_rows", r)
    e.solve_p_jacobi(10)
    t = [1e3 * e.time_jacobi(2000) for _ in range(3)]
    print("n=%d jacobi_tb_rows=%d: %s us/sweep" % (n, r, " ".join("%.3f" % x for x in t)), flush=True)
    e.close()
