#!/usr/bin/env python3
"""k_tm + k_jacobi_pair forced on the shipped-size fixtures: 128^2 dam-break (BASELINE configs[0]) to step 1000 and the
reference's 200^2 runs (fp64 and fp32), against the committed golden arrays.   python3 tools/probes/tm_golden.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "taichi-2d-vof_amd"))
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc
api = hip_api()
def run(fix, n, dtype, ic, steps_keys):
    z = np.load(os.path.join(ROOT, "tests", "golden", fix))
    e = Engine(api, make_desc(api, n, n, dtype, "f32", device=0))
    e.set_param("overlap_halves", 0); e.set_param("fuse_tm", 1); e.set_param("jacobi_pair", 2)
    e.set_init_F(ic)
    for st in steps_keys:
        e.step(st - e.istep)
        bad = [f for f in ("F", "u", "v", "p") if ("%s_%d" % (f, st)) in z and not np.array_equal(e.get(f), z["%s_%d" % (f, st)])]
        print("%s %d^2 %s step %d: %s  (tm_steps %d, pair_launches %d)" % (fix, n, dtype, st, "EQUAL" if not bad else "DIFFER " + ",".join(bad),
              e.get_counter("tm_steps"), e.get_counter("pair_launches")), flush=True)
run("dam128_f64.npz", 128, "f64", 1, (100, 1000))
