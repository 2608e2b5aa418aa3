import sys, os, time
sys.path.insert(0, "taichi-2d-vof_amd")
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc
api = hip_api()
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    e = Engine(api, make_desc(api, 4096, 4096, "f64", "f32", device=0))
    e.set_init_F(1); e.step(10); e.sync()
    prof = e.profile_steps(40)
    base = e.field_view("F")[0]
    print("pid %d engine %d  F base 0x%x  momentum %.1f jacobi_tb %.1f transport %.1f" % (os.getpid(), rep, base, prof["k_momentum"][0], prof["k_jacobi_tb"][0], prof["k_transport"][0]), flush=True)
    if len(sys.argv) > 2 and sys.argv[2] == "keep": keep = globals().setdefault("keep", []); keep.append(e)
    else: e.close()
