#!/usr/bin/env python3
"""Same-box A/B of a tuning knob over a 1000-step run (blocks of 100 steps): ms/step per block.
    python3 tools/adapt_ab.py [knob=jacobi_tb_adapt] [values=0,1] [n=4096] [steps=1000] [dtype=f64] [ic=1]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "taichi-2d-vof_amd"))
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc
knob = sys.argv[1] if len(sys.argv) > 1 else "jacobi_tb_adapt"
values = [float(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,1").split(",")]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
dtype = sys.argv[5] if len(sys.argv) > 5 else "f64"
ic = int(sys.argv[6]) if len(sys.argv) > 6 else 1
api = hip_api()
for rep in range(2):
    for v in values:
        e = Engine(api, make_desc(api, n, n, dtype, "f32", device=0))
        e.set_param(knob, v)
        e.set_init_F(ic)
        e.sync()
        out = []
        for _ in range(steps // 100):
            t0 = time.perf_counter(); e.step(100); e.sync()
            out.append(1e3 * (time.perf_counter() - t0) / 100)
        print("%s=%g: %s  mean %.4f" % (knob, v, " ".join("%.3f" % b for b in out), sum(out) / len(out)), flush=True)
        e.close()
