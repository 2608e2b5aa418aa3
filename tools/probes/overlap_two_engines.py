#!/usr/bin/env python3
"""How much would kernels of one step gain from overlapping each other's tails?  Two independent engines of
nx/2 x ny cells, each on its own stream, stepped (a) one after the other, (b) concurrently, against (c) one engine of
nx x ny.  If (b) is clearly faster than (c) for the same number of cells, a step split into two row halves whose
kernels interleave (the next kernel of the upper half filling the tail of the lower half's) is worth building.

    python3 tools/probes/overlap_two_engines.py [n=4096] [steps=200] [dtype=f64]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "taichi-2d-vof_amd"))
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dtype = sys.argv[3] if len(sys.argv) > 3 else "f64"
api = hip_api()
def mk(nx, ny):
    e = Engine(api, make_desc(api, nx, ny, dtype, "f32", device=0))
    e.set_init_F(1)
    e.step(24); e.sync()
    return e
def run(engs, k):
    for e in engs: e.sync()
    t0 = time.perf_counter()
    for c in range(k // 8):          # (8 steps = one graph launch per engine: the engines' launches alternate)
        for e in engs: e.step(8)
    for e in engs: e.sync()
    return 1e3 * (time.perf_counter() - t0) / k
parts = int(sys.argv[4]) if len(sys.argv) > 4 else 2
full = mk(n, n)
es = [mk(n // parts, n) for k in range(parts)]
for rep in range(3):
    tf = run([full], steps)
    alone = [run([e], steps) for e in es]
    tab = run(es, steps)
    print("full %dx%d: %.4f ms/step | %d parts alone: sum %.4f | concurrently: %.4f ms per step of all (%.1f %% of the full grid's)"
          % (n, n, tf, parts, sum(alone), tab, 100 * tab / tf), flush=True)
