#!/usr/bin/env python3
"""Per-kernel times (built-in profiler) and graph-replayed ms/step for a list of chunk-length settings.
    python3 tools/sweep_rows.py fctx_corr_rows 12,16,20,24,32 [n=4096] [dtype=f64] [ic=1]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "taichi-2d-vof_amd"))
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc
knob = sys.argv[1]
values = [float(v) for v in sys.argv[2].split(",")]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
dtype = sys.argv[4] if len(sys.argv) > 4 else "f64"
ic = int(sys.argv[5]) if len(sys.argv) > 5 else 1
api = hip_api()
for v in values + values[:1]:
    e = Engine(api, make_desc(api, n, n, dtype, "f32", device=0))
    e.set_param(knob, v)
    e.set_init_F(ic)
    e.step(10); e.sync()
    t0 = time.perf_counter(); e.step(40); e.sync()
    ms = 1e3 * (time.perf_counter() - t0) / 40
    prof = e.profile_steps(10)
    print("%s=%g: %.4f ms/step | %s" % (knob, v, ms, "  ".join("%s %.1f" % (k[2:], us) for k, (us, c) in sorted(prof.items()))), flush=True)
    e.close()
