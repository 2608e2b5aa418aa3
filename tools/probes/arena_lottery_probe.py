#!/usr/bin/env python3
"""Does a cheap probe at creation predict which of k_momentum's speeds an engine instance got?  For a series of
engines (half of them kept alive so that later arenas land elsewhere): k_momentum on the freshly zeroed fields
(3 profiled steps: F = 0 everywhere, the memory pattern of the real thing) against k_momentum of the dam-break run."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "taichi-2d-vof_amd"))
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc
api = hip_api()
keep = []
for k in range(14):
    e = Engine(api, make_desc(api, 4096, 4096, "f64", "f32", device=0))
    e.step(2); e.sync()
    z = e.profile_steps(6)
    e.set_init_F(1); e.step(20); e.sync()
    p = e.profile_steps(20)
    print("engine %2d  F @ 0x%x | zeros: momentum %.1f jacobi_tb %.1f transport %.1f | dam-break: momentum %.1f jacobi_tb %.1f transport %.1f" % (
        k, e.field_view("F")[0], z["k_momentum"][0], z["k_jacobi_tb"][0], z["k_transport"][0], p["k_momentum"][0], p["k_jacobi_tb"][0], p["k_transport"][0]), flush=True)
    if k % 2 == 0: keep.append(e)
    else: e.close()
