#!/usr/bin/env python3
"""How much do engine instances of ONE process differ in the k_tm form, and is an instance's speed its own?  (Round 6:
processes of one binary differ by up to 4 % on one box -- profiles/r06_compile_flags_ab_4_rcprio_reps.txt.)  Engines are
created one after the other (odd ones closed again, so that later arenas land in their holes), each timed twice over
steps 60-260 of a dam-break run; then the survivors once more, in reverse order."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "taichi-2d-vof_amd"))
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc
api = hip_api()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096


def timed(e, steps=200):
    e.sync(); t0 = time.perf_counter(); e.step(steps); e.sync()
    return 1e3 * (time.perf_counter() - t0) / steps


keep = []
if os.environ.get("SPREAD_DUMMY_FIRST"):     # something else takes the process's first big allocation
    dummy = Engine(api, make_desc(api, int(os.environ["SPREAD_DUMMY_FIRST"]), int(os.environ["SPREAD_DUMMY_FIRST"]), "f64", "f32", device=0))
    dummy.set_init_F(1); dummy.step(2); dummy.sync()
    if os.environ.get("SPREAD_DUMMY_FREE"):
        dummy.close()
for k in range(int(os.environ.get("SPREAD_ENGINES", "10"))):
    e = Engine(api, make_desc(api, n, n, "f64", "f32", device=0))
    e.set_init_F(1); e.step(60); e.sync()
    a, b = timed(e), timed(e)
    print("engine %2d  F @ 0x%x  ms/step %.4f %.4f" % (k, e.field_view("F")[0], a, b), flush=True)
    if k % 2 == 0: keep.append((k, e))
    else: e.close()
for k, e in reversed(keep):
    print("engine %2d again: %.4f %.4f (steps %d-)" % (k, timed(e), timed(e), e.istep - 400), flush=True)
