import os, sys, time
sys.path.insert(0, "taichi-2d-vof_amd")
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc
api = hip_api()
for halves in (0, 1, 0, 1):
    e = Engine(api, make_desc(api, 4096, 4096, "f64", "f32", device=0))
    e.set_param("overlap_halves", halves)
    e.set_init_F(1); e.step(1); e.sync()
    ts = []
    for k in range(4):
        t0 = time.perf_counter(); e.step(8); t1 = time.perf_counter(); e.sync(); t2 = time.perf_counter()
        ts.append("%.2f+%.2f" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)))
    print("halves", halves, "step(8) host ms + sync ms:", " ".join(ts), flush=True)
    e.close()
