#!/usr/bin/env python3
"""Debug: the chained k_tm batches against the plain sequence at 4096^2 -- where do u / v in memory differ?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "taichi-2d-vof_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc

api = hip_api()
n = 4096
def eng(**p):
    e = Engine(api, make_desc(api, n, n, "f64", "f32", device=0))
    for k, v in p.items():
        e.set_param(k, v)
    e.set_init_F(1)
    return e
off = eng(jacobi_tb_adapt=0, overlap_halves=0, fuse_tm=0)
cases = {"default": eng(), "plan off": eng(jacobi_tb_adapt=0), "no pair kernel": eng(jacobi_pair=0)}
for st in (70, 100, 130):
    off.step(st - off.istep)
    ref = {f: off.get(f) for f in ("u", "v", "p")}
    for name, e in cases.items():
        while e.istep < st:
            e.step(min(16 if "16" in name else 10, st - e.istep))
        for f in ("u", "v", "p"):
            x = e.get(f)
            bad = np.argwhere(x != ref[f])
            if len(bad) == 0:
                print("step %d %-26s %s: equal" % (st, name, f))
                continue
            rows = np.unique(bad[:, 0])
            rel = np.abs(x - ref[f])[x != ref[f]] / np.maximum(np.abs(ref[f][x != ref[f]]), 1e-300)
            print("step %d %-26s %s: %d cells differ; rows %d..%d (%d distinct), cols %d..%d; relative difference median %.2e max %.2e" % (
                st, name, f, len(bad), rows.min(), rows.max(), len(rows), bad[:, 1].min(), bad[:, 1].max(), np.median(rel), rel.max()))
