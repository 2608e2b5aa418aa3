#!/usr/bin/env python3
"""overlap_halves x jacobi_tb_adapt at a late phase of the 4096^2 dam-break (steps skip .. skip + 600).
    python3 tools/probes/halves_late.py [skip=640]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "taichi-2d-vof_amd"))
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc
skip = int(sys.argv[1]) if len(sys.argv) > 1 else 640
api = hip_api()
def mk(halves, adapt):
    e = Engine(api, make_desc(api, 4096, 4096, "f64", "f32", device=0))
    e.set_param("overlap_halves", halves); e.set_param("jacobi_tb_adapt", adapt)
    e.set_init_F(1); e.step(skip); e.sync(); return e
def run(e, k):
    e.sync(); t0 = time.perf_counter(); e.step(k); e.sync()
    return 1e3 * (time.perf_counter() - t0) / k
es = [("base", mk(0, 1)), ("base/noplan", mk(0, 0)), ("halves", mk(1, 1)), ("halves/noplan", mk(1, 0))]
for r in range(3):
    print("from step %d: " % (skip + 200 * r) + "  ".join("%s %.4f (plan %d)" % (n, run(e, 200), e.get_counter("tb_plan_active")) for n, e in es), flush=True)
