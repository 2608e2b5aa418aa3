#!/usr/bin/env python3
"""k_tm against the plain sequence on a spread of mid-size grids (the store-data hazard of round 4 only showed from 1024^2 up
and not in every run): equality of the state after 43 steps, each configuration twice.   python3 tools/probes/tm_stress.py"""
import os, sys, itertools
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "taichi-2d-vof_amd"))
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc
api = hip_api()
cases = [(2048, 2048, "f64", 3, 0), (3072, 3072, "f64", 2, 0), (1280, 1280, "f64", 1, 0), (1024, 1024, "f64", 1, 0), (1536, 2048, "f64", 2, 0), (2048, 1025, "f64", 3, 0), (3072, 3072, "f64", 1, 64), (2048, 2048, "f32", 1, 0),
         (2500, 1800, "f32", 2, 48), (1111, 2222, "f64", 1, 37), (4096, 1024, "f64", 3, 0), (1024, 4096, "f32", 3, 16), (2048, 3000, "f64", 2, 0)]
bad = 0
for nx, ny, dtype, ic, rows in cases:
    for rep in range(2):
        es = []
        for tm in (1, 0):
            e = Engine(api, make_desc(api, nx, ny, dtype, "f32", device=0))
            e.set_param("overlap_halves", 0); e.set_param("fuse_tm", tm); e.set_param("tm_rows", rows); e.set_param("jacobi_pair", 2 if tm else 0)
            e.set_init_F(ic); e.step(43); es.append(e)
        diff = [f for f in ("F", "u", "v", "p", "u_star", "v_star", "rhs") if not np.array_equal(es[0].get(f), es[1].get(f))]
        print("%dx%d %s ic %d rows %d rep %d: %s" % (nx, ny, dtype, ic, rows, rep, "EQUAL" if not diff else "DIFFER " + ",".join(diff)), flush=True)
        bad += bool(diff)
        for e in es: e.close()
print("mismatches:", bad)
