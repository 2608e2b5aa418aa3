#!/usr/bin/env python3
"""Wall time per step in consecutive blocks of a long run (clock / thermal drift check).
    python tools/sustained.py [--nx 4096 --ny 4096 --block 50 --blocks 40 --sleep-after 20]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "taichi-2d-vof_amd"))
ap = argparse.ArgumentParser()
ap.add_argument("--nx", type=int, default=4096); ap.add_argument("--ny", type=int, default=4096)
ap.add_argument("--block", type=int, default=50); ap.add_argument("--blocks", type=int, default=40)
ap.add_argument("--sleep-after", type=int, default=20, help="sleep 2 s after this block index")
a = ap.parse_args()
from vof2d._lib import hip_api
from vof2d.engine import Engine, make_desc
api = hip_api()
e = Engine(api, make_desc(api, a.nx, a.ny, "f64", "f32", device=0))
e.set_init_F(1); e.step(5); e.sync()
out = []
for b in range(a.blocks):
    if b == a.sleep_after:
        time.sleep(2.0); out.append("|sleep|")
    t0 = time.perf_counter(); e.step(a.block); e.sync()
    out.append("%.0f" % (1e6 * (time.perf_counter() - t0) / a.block))
print("us/step per block of %d steps:" % a.block, " ".join(out))
