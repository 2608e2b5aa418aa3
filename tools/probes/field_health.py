#!/usr/bin/env python3
"""NaN / overflow / tiny-value census of the fields during a long run.
    python tools/probes/field_health.py N dt step,step,...     (e.g. 8192 1e-6 10,200,1000)"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "taichi-2d-vof_amd"))
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc
api = hip_api()
n, dt, pts = int(sys.argv[1]), float(sys.argv[2]), [int(x) for x in sys.argv[3].split(",")]
e = Engine(api, make_desc(api, n, n, "f64", "f32", device=0, dt=dt)); e.set_init_F(1); done = 0
for t in pts:
    e.step(t - done); done = t
    s = []
    for f in ("F", "u", "v", "p"):
        x = e.get(f)
        ax = np.abs(x[np.isfinite(x)])
        s.append("%s nan %d max %.3g tiny %d" % (f, np.isnan(x).sum(), ax.max() if ax.size else float("nan"),
                                                 ((ax < 1e-280) & (ax > 0)).sum()))
    print(n, "dt", dt, "step", t, "courant", e.get_counter("courant_violations"), " | ".join(s), flush=True)
