#!/usr/bin/env python3
"""overlap_halves on / off in one process: wall ms/step (alternating runs) and equality of the state.
    python3 tools/probes/halves_ab.py [n=4096] [steps=200] [dtype=f64] [ic=1]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "taichi-2d-vof_amd"))
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dtype = sys.argv[3] if len(sys.argv) > 3 else "f64"
ic = int(sys.argv[4]) if len(sys.argv) > 4 else 1
extra = dict(kv.split("=") for kv in sys.argv[5:])
api = hip_api()
def mk(halves, adapt=1):
    e = Engine(api, make_desc(api, n, n, dtype, "f32", device=0))
    e.set_param("overlap_halves", halves)
    e.set_param("jacobi_tb_adapt", adapt)
    for k, v in extra.items():
        if halves: e.set_param(k, float(v))
    e.set_init_F(ic)
    e.step(24); e.sync()
    return e
def run(e, k):
    e.sync(); t0 = time.perf_counter(); e.step(k); e.sync()
    return 1e3 * (time.perf_counter() - t0) / k
engs = [("base", mk(0)), ("base, no plan", mk(0, 0)), ("halves", mk(1))]
for rep in range(4):
    print("  ".join("%s %.4f" % (name, run(e, steps)) for name, e in engs), "ms/step", flush=True)
ref = engs[0][1]
for name, e in engs[1:]:
    same = all(np.array_equal(ref.get(f), e.get(f)) for f in ("F", "u", "v", "p"))
    print("%s == base after %d steps: %s (courant %s / %s)" % (name, 24 + 4 * steps, same, e.get_counter("courant_violations"), ref.get_counter("courant_violations")))
