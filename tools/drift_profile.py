#!/usr/bin/env python3
"""Per-kernel times (in-library profiler) at several points of a long run.
    python tools/drift_profile.py [--nx 8192 --ny 8192 --at 10,200,600,1200,2000]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "taichi-2d-vof_amd"))
ap = argparse.ArgumentParser()
ap.add_argument("--nx", type=int, default=8192); ap.add_argument("--ny", type=int, default=8192)
ap.add_argument("--at", default="10,200,600,1200,2000"); ap.add_argument("-ic", type=int, default=1)
ap.add_argument("--set", default="", help="knob=value[,knob=value] tuning parameters (vof_set_param)")
a = ap.parse_args()
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc
api = hip_api()
e = Engine(api, make_desc(api, a.nx, a.ny, "f64", "f32", device=0))
for kv in [x for x in a.set.split(",") if x]:
    e.set_param(kv.split("=")[0], float(kv.split("=")[1]))
e.set_init_F(a.ic)
done = 0
for tgt in [int(x) for x in a.at.split(",")]:
    e.step(max(0, tgt - done)); e.sync(); done = max(done, tgt)
    t0 = time.perf_counter(); e.step(10); e.sync(); w = 1e5 * (time.perf_counter() - t0); done += 10
    prof = e.profile_steps(4); done += 4
    print("step %5d: wall %7.1f us/step | " % (tgt, w) + "  ".join("%s %.0f" % (k[2:], us) for k, (us, n) in sorted(prof.items())), flush=True)
