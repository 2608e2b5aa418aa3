#!/usr/bin/env python3
"""Run-to-run bimodality of the step time: fresh engines in one process, steady-state ms/step
(steps 700-1000) against the device address of the field arena."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "taichi-2d-vof_amd"))
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc
api = hip_api()
keep = []
for rep in range(8):
    e = Engine(api, make_desc(api, 4096, 4096, "f64", "f32", device=0))
    e.set_init_F(1)
    e.step(700); e.sync()
    t0 = time.perf_counter(); e.step(300); e.sync()
    ms = 1e3 * (time.perf_counter() - t0) / 300
    bases = {f: e.field_view(f)[0] for f in ("F", "u", "p", "rhs")}
    print("engine %d: %.4f ms/step  F @ 0x%x (mod 2MiB %d KiB, mod 1GiB %d MiB)  p-F %d" % (
        rep, ms, bases["F"], (bases["F"] % (2 << 20)) >> 10, (bases["F"] % (1 << 30)) >> 20, bases["p"] - bases["F"]), flush=True)
    if rep % 2 == 0:
        keep.append(e)       # keep some alive so that later arenas land elsewhere
    else:
        e.close()
