#!/usr/bin/env python3
"""fuse_tm on / off: equality of the state and wall ms/step.   python3 tools/probes/tm_check.py [nx] [ny] [dtype] [ic] [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "taichi-2d-vof_amd"))
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 448
ny = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dtype = sys.argv[3] if len(sys.argv) > 3 else "f64"
ic = int(sys.argv[4]) if len(sys.argv) > 4 else 1
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 0
api = hip_api()
def mk(tm, **kw):
    e = Engine(api, make_desc(api, nx, ny, dtype, "f32", device=0))
    e.set_param("overlap_halves", 0)
    e.set_param("fuse_tm", tm)
    for k, v in kw.items(): e.set_param(k, v)
    e.set_init_F(ic)
    return e
kw = dict((k, float(v)) for k, v in (x.split('=') for x in sys.argv[6:]))
a, b = mk(1, **kw), mk(0)
for st in (1, 3, 11, 12, 22, 23, 40, 41, 57):
    a.step(st - a.istep); b.step(st - b.istep)
    bad = []
    for f in ("F", "u", "v", "p", "u_star", "v_star", "rhs"):
        x, y = a.get(f), b.get(f)
        if not np.array_equal(x, y):
            d = np.argwhere(x != y)
            bad.append("%s: %d cells, rows %d..%d cols %d..%d, first %s %r vs %r" % (f, len(d), d[:, 0].min(), d[:, 0].max(), d[:, 1].min(), d[:, 1].max(), d[0], x[tuple(d[0])], y[tuple(d[0])]))
    print("step %d: %s (courant %d / %d)" % (st, "EQUAL" if not bad else " | ".join(bad), a.get_counter("courant_violations"), b.get_counter("courant_violations")), flush=True)
    if bad: break
if steps:
    def run(e, k):
        e.sync(); t0 = time.perf_counter(); e.step(k); e.sync(); return 1e3 * (time.perf_counter() - t0) / k
    for r in range(3):
        print("ms/step: fused %.4f  base %.4f" % (run(a, steps), run(b, steps)), flush=True)
