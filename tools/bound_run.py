#!/usr/bin/env python3
"""The workload behind profiles/<tag>_bound.md: N fused steps from set_init_F, nothing else in the process
(no torch, no timing legs), so that every dispatch of a rocprofv3 --pmc pass belongs to a known step:
the n-th k_momentum dispatch IS step n.

    python3 tools/bound_run.py [--nx 4096] [--dtype f64] [-ic 1] [--steps 360] [--lib PATH]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "taichi-2d-vof_amd"))
ap = argparse.ArgumentParser()
ap.add_argument("--nx", type=int, default=4096)
ap.add_argument("--ny", type=int, default=0)
ap.add_argument("--dtype", default="f64")
ap.add_argument("-ic", type=int, default=1)
ap.add_argument("--steps", type=int, default=360)
ap.add_argument("--dt", type=float, default=0.0)
ap.add_argument("--lib", default="")
ap.add_argument("--param", action="append", default=[])
a = ap.parse_args()
from vof2d import _abi
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc
api = _abi.bind(ctypes.CDLL(a.lib, mode=ctypes.RTLD_GLOBAL), "vof_") if a.lib else hip_api()
kw = {"dt": a.dt} if a.dt > 0 else {}
e = Engine(api, make_desc(api, a.nx, a.ny or a.nx, a.dtype, "f32", device=0, **kw))
for k, v in (kv.split("=") for kv in a.param):
    e.set_param(k, float(v))
e.set_init_F(a.ic)
e.step(a.steps)
e.sync()
print("ran", a.steps, "steps of", a.nx, a.ny or a.nx, a.dtype, "ic", a.ic)
