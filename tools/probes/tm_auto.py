#!/usr/bin/env python3
"""The default (fuse_tm = -1) at work: which form stays, and ms/step before / after.  python3 tools/probes/tm_auto.py [n] [ic] [dtype]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "taichi-2d-vof_amd"))
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ic = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dtype = sys.argv[3] if len(sys.argv) > 3 else "f64"
api = hip_api()
kw = {"dt": 1e-6} if n > 4096 else {}
e = Engine(api, make_desc(api, n, n, dtype, "f32", device=0, **kw))
e.set_init_F(ic)
for k in range(8):
    e.sync(); t0 = time.perf_counter(); e.step(40); e.sync()
    print("%dx%d ic %d %s steps %3d-%3d: %.4f ms/step  tm_choice %d  tm_steps %d  halves_steps %d" % (n, n, ic, dtype, 40 * k + 1, 40 * k + 40, 1e3 * (time.perf_counter() - t0) / 40,
          e.get_counter("tm_choice"), e.get_counter("tm_steps"), e.get_counter("halves_steps")), flush=True)
