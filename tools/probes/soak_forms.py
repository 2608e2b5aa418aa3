#!/usr/bin/env python3
"""Long runs of every batch form side by side on the same data: the default (rule), the pair kernels forced (k_tm + k_jacobi_pair),
the chains forced, the plain four-kernel sequence -- equality of F, u, v, p at several points of runs of thousands of steps
(intermittent hazards show in long runs at size, not in the short parity tests: the store-data hazard of round 4 did).

    python3 tools/probes/soak_forms.py [--reps 2]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "taichi-2d-vof_amd"))
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()
api = hip_api()
FORMS = (("default", {}), ("pairs", {"overlap_halves": 0, "fuse_tm": 1, "jacobi_pair": 2}), ("chains", {"overlap_halves": 2, "fuse_tm": 0}),
         ("plain", {"overlap_halves": 0, "fuse_tm": 0}))
CASES = ((4096, "f64", 1, (1000, 3000, 6000)), (4096, "f32", 1, (1000, 3000)), (3072, "f64", 2, (500, 2000)), (2048, "f32", 3, (1000, 4000)),
         (2560, "f64", 3, (700, 2500)))
bad_total = 0
for rep in range(a.reps):
    for n, dtype, ic, points in CASES:
        es = []
        for name, knobs in FORMS:
            e = Engine(api, make_desc(api, n, n, dtype, "f32", device=0))
            for k, v in knobs.items():
                e.set_param(k, v)
            e.set_init_F(ic)
            es.append((name, e))
        for st in points:
            for _, e in es:
                e.step(st - e.istep)
            ref = {f: es[-1][1].get(f) for f in ("F", "u", "v", "p")}
            bad = [(name, f) for name, e in es[:-1] for f in ref if not np.array_equal(e.get(f), ref[f])]
            bad_total += len(bad)
            print("rep %d  %d^2 %s ic %d step %d: %s | tm_steps %s pair_launches %s halves_steps %s courant %s" % (
                rep, n, dtype, ic, st, "ALL EQUAL" if not bad else "DIFFER %r" % bad, [e.get_counter("tm_steps") for _, e in es],
                [e.get_counter("pair_launches") for _, e in es], [e.get_counter("halves_steps") for _, e in es],
                [e.get_counter("courant_violations") for _, e in es]), flush=True)
        for _, e in es:
            e.close()
print("soak: %s" % ("OK" if bad_total == 0 else "%d DIFFERENCES" % bad_total))
sys.exit(1 if bad_total else 0)
