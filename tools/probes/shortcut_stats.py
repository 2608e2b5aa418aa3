#!/usr/bin/env python3
"""How often the exact wave-level shortcuts of k_momentum / k_transport are taken, per configuration
(library built with -DVOF_SHORTCUT_STATS: make -C taichi-2d-vof_amd/csrc variant NAME=stats EXTRA=-DVOF_SHORTCUT_STATS).
One process per configuration (the counters are per process): steps 101-200 of each."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "taichi-2d-vof_amd"))
if len(sys.argv) > 1:
    n, dtype, ic, steps = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    from vof2d import _abi
    from vof2d.engine import Engine, make_desc
    api = _abi.bind(ctypes.CDLL(os.path.join(ROOT, "taichi-2d-vof_amd/csrc/build/variants/libvof2d_stats.so"), mode=ctypes.RTLD_GLOBAL), "vof_")
    e = Engine(api, make_desc(api, n, n, dtype, "f32", device=0))
    e.set_init_F(ic); e.step(steps); e.sync(); e.close()
else:
    for cfg in (("4096", "f64", "1", "300"), ("2048", "f32", "2", "300"), ("2048", "f64", "3", "300")):
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + list(cfg), capture_output=True, text=True)
        print("%s^2 %s -ic %s, steps 1-%s:" % cfg, [l for l in r.stderr.splitlines() if "shortcut stats" in l][-1:] or r.stderr[-300:])
