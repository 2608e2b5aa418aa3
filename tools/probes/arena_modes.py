#!/usr/bin/env python3
"""Does the placement of the field arena decide which of k_momentum's two speeds an engine gets (164 / 181 us with
buffer stores, 175 / 188 without; the other kernels do not care)?  Engines of a library built with -DVOF_ARENA_EXP,
one after the other in ONE process: shift of the whole arena inside its allocation, extra bytes between fields.

    make -C taichi-2d-vof_amd/csrc variant NAME=arena EXTRA=-DVOF_ARENA_EXP && python3 tools/probes/arena_modes.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "taichi-2d-vof_amd"))
from vof2d import _abi
from vof2d.engine import Engine, make_desc
api = _abi.bind(ctypes.CDLL(os.path.join(ROOT, "taichi-2d-vof_amd/csrc/build/variants/libvof2d_arena.so"), mode=ctypes.RTLD_GLOBAL), "vof_")
keep = []
def one(shift, skew, hold=False):
    os.environ["VOF2D_ARENA_SHIFT"], os.environ["VOF2D_FIELD_SKEW"] = str(shift), str(skew)
    e = Engine(api, make_desc(api, 4096, 4096, "f64", "f32", device=0))
    e.set_init_F(1); e.step(30); e.sync()
    p = e.profile_steps(30)
    base = e.field_view("F")[0]
    print("shift %9d skew %9d  F @ 0x%x (mod 2 MiB: %7d)  momentum %.1f  jacobi_tb %.1f  transport %.1f" % (
        shift, skew, base, base % (2 << 20), p["k_momentum"][0], p["k_jacobi_tb"][0], p["k_transport"][0]), flush=True)
    if hold: keep.append(e)
    else: e.close()
for rep in range(2):
    for shift in (0, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 16 << 20, 256 << 20):
        one(shift, 0)
for skew in (0, 128, 4096, 65536, 1 << 20, 2 << 20, 5 << 20, 32 << 20):
    one(0, skew)
for k in range(6):
    one(0, 0, hold=True)
