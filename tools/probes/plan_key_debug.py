#!/usr/bin/env python3
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "taichi-2d-vof_amd"))
from vof2d import _abi
from vof2d.engine import Engine, make_desc
lib = C.CDLL(os.path.join(ROOT, "taichi-2d-vof_amd", "csrc", "build", "variants", "libvof2d_wavetimes.so"))
api = _abi.bind(lib, "vof_")
e = Engine(api, make_desc(api, 4096, 4096, "f64", "f32", device=0))
e.set_param("fuse_tm", 1)
e.set_init_F(1)
for st in (64, 96, 97, 128):
    e.step(st - e.istep)
    w = e.get_counter("dbg_plan_word")
    print("step %d: plan word %#x -> active %d waves %d R %d ntt %d | a launch expects waves %d R %d ntt %d" % (
        st, w, w & 1, (w >> 1) & 0xffffffff, (w >> 33) & 0x7fff, w >> 48, e.get_counter("dbg_plan_waves"), e.get_counter("dbg_plan_R"), e.get_counter("dbg_plan_ntt")))
