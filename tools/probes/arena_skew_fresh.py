#!/usr/bin/env python3
"""k_momentum against the spacing of the fields, ONE engine per fresh process (the first arena of a process is
reliably of the slow kind): library built with -DVOF_ARENA_EXP (VOF2D_FIELD_SKEW = extra bytes between fields)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "taichi-2d-vof_amd"))
if len(sys.argv) > 1:
    from vof2d import _abi
    from vof2d.engine import Engine, make_desc
    api = _abi.bind(ctypes.CDLL(os.path.join(ROOT, "taichi-2d-vof_amd/csrc/build/variants/libvof2d_arena.so"), mode=ctypes.RTLD_GLOBAL), "vof_")
    e = Engine(api, make_desc(api, 4096, 4096, "f64", "f32", device=0))
    e.set_init_F(1); e.step(30); e.sync()
    p = e.profile_steps(30)
    print("skew %10s shift %8s: momentum %.1f jacobi_tb %.1f transport %.1f" % (os.environ.get("VOF2D_FIELD_SKEW"), os.environ.get("VOF2D_ARENA_SHIFT"), p["k_momentum"][0], p["k_jacobi_tb"][0], p["k_transport"][0]))
else:
    for rep in range(2):
        for skew in (0, 4096, 65536, 1 << 20, (2 << 20) + 4096, 5 << 20, (8 << 20) + 65536, 32 << 20, (33 << 20) + 4096, 67 << 20):
            env = dict(os.environ, VOF2D_FIELD_SKEW=str(skew), VOF2D_ARENA_SHIFT="0")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "x"], capture_output=True, text=True, env=env)
            print(r.stdout.strip() or r.stderr[-200:], flush=True)
