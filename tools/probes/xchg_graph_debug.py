#!/usr/bin/env python3
"""Does this RCCL survive being captured into a hipGraph?  Loopback strip, few steps, verbose."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "taichi-2d-vof_amd"))
os.environ["VOF2D_DEBUG"] = "1"
from vof2d import _abi
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc, comm_unique_id
api = hip_api()
W = _abi.halo_rows(10)
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
e = Engine(api, make_desc(api, 512, 512, "f64", "f32", rows=(200 - W, 300 + W), own=(200, 300), device=0))
e.set_init_F(1)
e.comm_init(comm_unique_id(api), 0, 1, loopback=True)
for k in range(6):
    print("step", k, "mode", mode, flush=True)
    e.step_exchange(1, mode); e.sync()
    print("   done; graph steps so far:", e.get_counter("exchange_graph_steps"), flush=True)
e.comm_destroy()
print("OK")
