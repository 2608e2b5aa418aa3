import sys, os
sys.path[:0] = ["/root/repo/taichi-2d-vof_amd", "/root/repo/tests"]
import numpy as np
from vof2d._lib import hip_api
from util import engine, STATE
api = hip_api()
for n, R in ((300, 6), (300, 5), (300, 7), (640, 6), (300, 16)):
    a = engine(api, n, n, "f64", "f32", ic=1); a.set_param("fuse_tm", 1); a.set_param("tm_rows", R); a.set_param("overlap_halves", 0)
    b = engine(api, n, n, "f64", "f32", ic=1); b.set_param("fuse_tm", 0); b.set_param("overlap_halves", 0)
    for st in (11, 41):
        a.step(st - a.istep); b.step(st - b.istep)
        bad = [f for f in STATE if not np.array_equal(a.get(f), b.get(f))]
        print(n, R, st, "tm_steps", a.get_counter("tm_steps"), "differs:", bad)
