#!/usr/bin/env python3
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "taichi-2d-vof_amd"))
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc
nx, ny, dtype = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
kw = dict((k, float(v)) for k, v in (x.split('=') for x in sys.argv[4:]))
api = hip_api()
def mk(tm):
    e = Engine(api, make_desc(api, nx, ny, dtype, "f32", device=0))
    e.set_param("overlap_halves", 0); e.set_param("fuse_tm", tm)
    for k, v in kw.items(): e.set_param(k, v)
    e.set_init_F(1); return e
a, a2, b = mk(1), mk(1), mk(0)
for e in (a, a2, b): e.step(3)
def cmp(x, y, tag):
    out = []
    for f in ("F", "u", "v", "p", "u_star", "v_star", "rhs"):
        X, Y = x.get(f), y.get(f)
        d = np.argwhere(X != Y)
        if len(d): out.append("%s %d cells rows %d..%d cols %d..%d" % (f, len(d), d[:, 0].min(), d[:, 0].max(), d[:, 1].min(), d[:, 1].max()))
    print("%dx%d %s %s %s: %s" % (nx, ny, dtype, kw, tag, "EQUAL" if not out else " | ".join(out)), flush=True)
cmp(a, b, "fused vs base"); cmp(a, a2, "fused vs fused")
