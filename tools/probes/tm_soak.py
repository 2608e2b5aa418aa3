#!/usr/bin/env python3
"""Long run of the default (forms timed again every 16 batches) against the plain four-kernel sequence: equality of
the state at steps 1000, 3000, 6000 (4096^2 fp64 dam-break; 3072^2 fp64 bubble).   python3 tools/probes/tm_soak.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "taichi-2d-vof_amd"))
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc
api = hip_api()
for n, ic in ((4096, 1), (3072, 2)):
    a = Engine(api, make_desc(api, n, n, "f64", "f32", device=0)); a.set_param("tune_period", 16); a.set_init_F(ic)
    b = Engine(api, make_desc(api, n, n, "f64", "f32", device=0)); b.set_param("overlap_halves", 0); b.set_param("fuse_tm", 0); b.set_init_F(ic)
    for st in (1000, 3000, 6000):
        a.step(st - a.istep); b.step(st - b.istep)
        bad = [f for f in ("F", "u", "v", "p") if not np.array_equal(a.get(f), b.get(f))]
        print("%d^2 ic %d step %d: %s | tm_steps %d halves_steps %d pair_launches %d tm_choice %d courant %d/%d" % (n, ic, st, "EQUAL" if not bad else "DIFFER " + ",".join(bad),
              a.get_counter("tm_steps"), a.get_counter("halves_steps"), a.get_counter("pair_launches"), a.get_counter("tm_choice"),
              a.get_counter("courant_violations"), b.get_counter("courant_violations")), flush=True)
    a.close(); b.close()
