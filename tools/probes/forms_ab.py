#!/usr/bin/env python3
"""ms/step of the batch forms of a full domain side by side (engines alive together, taking turns): the default (the rule of
vof_step), k_tm + k_jacobi_pair forced, chains / plain forced.
    python3 tools/probes/forms_ab.py [n=4096] [dtype=f64] [ic=1] [steps=200] [rounds=3] [ny=n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "taichi-2d-vof_amd"))
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dtype = sys.argv[2] if len(sys.argv) > 2 else "f64"
ic = int(sys.argv[3]) if len(sys.argv) > 3 else 1
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 200
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 3
ny = int(sys.argv[6]) if len(sys.argv) > 6 else n
api = hip_api()
kw = {"dt": 1e-6} if n > 4096 else {}
forms = (("default", {}), ("k_tm + pairs", {"fuse_tm": 1, "jacobi_pair": 2}), ("k_tm, two tb launches", {"fuse_tm": 1, "jacobi_pair": 0}), ("chains / plain", {"fuse_tm": 0}))
if dtype == "f32" and os.environ.get("VOF2D_FORMS_VEC4"):      # (needs the -DVOF_PAIR_VEC4 experiment build)
    forms += (("k_tm + pairs, 4 per lane", {"fuse_tm": 1, "jacobi_pair": 2, "pair_vec4": 1}), ("k_tm (4 per lane), tb", {"fuse_tm": 1, "jacobi_pair": 0, "pair_vec4": 1}))
engs = []
for name, knobs in forms:
    e = Engine(api, make_desc(api, n, ny, dtype, "f32", device=0, **kw))
    for k, v in knobs.items():
        e.set_param(k, v)
    e.set_init_F(ic)
    e.step(33)
    e.sync()
    engs.append((name, e, []))
for r in range(rounds):
    for name, e, acc in engs:
        e.sync(); t0 = time.perf_counter(); e.step(steps); e.sync()
        acc.append(1e3 * (time.perf_counter() - t0) / steps)
print("%d x %d %s ic %d, %d rounds of %d steps, ms/step:" % (n, ny, dtype, ic, rounds, steps))
for name, e, acc in engs:
    print("  %-24s %s   (tm_choice %d, gas share %.3f, tm_steps %d, pair launches %d, halves_steps %d)" % (
        name, " ".join("%.4f" % x for x in acc), e.get_counter("tm_choice"), e.get_param("gas_share"), e.get_counter("tm_steps"), e.get_counter("pair_launches"), e.get_counter("halves_steps")))
import hashlib
def digest(e):
    h = hashlib.sha256()
    for f in ("F", "u", "v", "p"):
        h.update((e.get(f) + 0.0).tobytes())
    return h.hexdigest()[:12]
print("  state after %d steps: %s" % (engs[0][1].istep, " ".join(digest(e) for _, e, _ in engs)))
