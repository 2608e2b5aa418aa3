#!/usr/bin/env python3
"""Would the NEXT step's k_jacobi_pair fit into the slots the tail of k_tm leaves empty?  Diagnostic build
(make -C taichi-2d-vof_amd/csrc wavetimes); timing only (vof_debug_time_overlap: the Jacobi launch reads the rhs the k_tm launch beside
it is writing; the state the steps run on is not touched).

    python3 tools/probes/overlap_tail.py [--n 4096] [--at 96,704] [--reps 12] [-ic 1]

us per [k_tm, k_jacobi_pair] of a step: one stream (what the step does) / k_tm at the highest stream priority beside k_jacobi_pair
at the lowest / two plain streams / the priorities the other way round."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "taichi-2d-vof_amd"))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--at", default="96,704")
ap.add_argument("--reps", type=int, default=12)
ap.add_argument("-ic", type=int, default=1)
ap.add_argument("--param", action="append", default=[])
a = ap.parse_args()
from vof2d import _abi
from vof2d.engine import Engine, make_desc

lib = C.CDLL(os.path.join(ROOT, "taichi-2d-vof_amd", "csrc", "build", "variants", "libvof2d_wavetimes.so"))
api = _abi.bind(lib, "vof_")
to = lib.vof_debug_time_overlap
to.restype = C.c_int
to.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_float)]
kw = {"dt": 1e-6} if a.n > 4096 else {}
e = Engine(api, make_desc(api, a.n, a.n, "f64", "f32", device=0, **kw))
e.set_param("fuse_tm", 1)
for kv in a.param:
    e.set_param(kv.split("=")[0], float(kv.split("=")[1]))
e.set_init_F(a.ic)
NAMES = {0: "one stream", 1: "k_tm high / pair low priority", 2: "two plain streams", 3: "k_tm low / pair high"}


def t(mode):
    us = C.c_float(0)
    rc = to(e._h, mode, a.reps, C.byref(us))
    assert rc == 0, (rc, mode)
    return us.value


done = 0
for at in [int(x) for x in a.at.split(",")]:
    e.step(at - done)
    done = at
    e.sync()
    t(0)
    res = {m: [] for m in NAMES}
    for rnd in range(3):
        for m in NAMES:
            res[m].append(t(m))
    print("== %d^2 fp64 ic %d after %d steps: us per [k_tm, k_jacobi_pair] (three rounds) | %s" % (
        a.n, a.ic, at, " | ".join("%s %s" % (NAMES[m], " / ".join("%.1f" % x for x in res[m])) for m in NAMES)), flush=True)
