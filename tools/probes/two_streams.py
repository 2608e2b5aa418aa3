#!/usr/bin/env python3
"""Do the tails of the step's kernels leave room?  One 4096x4096 engine against two concurrent
2048x4096 engines (own streams, same total cells): ms per step of the pair."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "taichi-2d-vof_amd"))
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc
api = hip_api()
def run(engs, n=60):
    for e in engs: e.set_init_F(1)
    for e in engs: e.step(10)
    for e in engs: e.sync()
    t0 = time.perf_counter()
    for k in range(n // 10):
        for e in engs: e.step(10)
    for e in engs: e.sync()
    return 1e3 * (time.perf_counter() - t0) / n
for rep in range(2):
    one = [Engine(api, make_desc(api, 4096, 4096, "f64", "f32", device=0))]
    print("one 4096x4096: %.4f ms/step" % run(one)); [e.close() for e in one]
    two = [Engine(api, make_desc(api, 2048, 4096, "f64", "f32", device=0)) for _ in range(2)]
    print("two 2048x4096 concurrently: %.4f ms/step (both)" % run(two)); [e.close() for e in two]
    half = [Engine(api, make_desc(api, 2048, 4096, "f64", "f32", device=0))]
    print("one 2048x4096 alone: %.4f ms/step" % run(half)); [e.close() for e in half]
