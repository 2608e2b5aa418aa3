#!/usr/bin/env python3
"""Where do the slow waves of a one-round launch sit?  Per-wave durations (diagnostic build: make -C taichi-2d-vof_amd/csrc
wavetimes) of k_momentum / k_transport as a (chunk row x tile column) map: means per chunk row and per tile column.
    python3 tools/probes/wave_map.py [nx=2048] [ny=2048] [dtype=f64] [ic=1] [at=40]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "taichi-2d-vof_amd"))
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ny = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
dtype = sys.argv[3] if len(sys.argv) > 3 else "f64"
ic = int(sys.argv[4]) if len(sys.argv) > 4 else 1
at = int(sys.argv[5]) if len(sys.argv) > 5 else 40
os.environ["VOF2D_OVERLAP_HALVES"] = "0"
from vof2d import _abi
from vof2d.engine import Engine, make_desc
lib = C.CDLL(os.path.join(ROOT, "taichi-2d-vof_amd", "csrc", "build", "variants", "libvof2d_wavetimes.so"))
api = _abi.bind(lib, "vof_")
dbg = lib.vof_debug_wave_times
dbg.restype = C.c_int; dbg.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32]
e = Engine(api, make_desc(api, nx, ny, dtype, "f32", device=0))
e.set_param("jacobi_tb_adapt", 0)      # (no planner block: wave index = 4 * block + wave in block)
e.set_init_F(ic)
e.step(at)
cap = 1 << 16
for name, kid, stride in (("k_momentum", 0, 124), ("k_transport", 12, 112)):
    assert dbg(e._h, kid, None, cap) == 0
    e.step(2)
    out = np.zeros((cap, 2), np.uint64)
    assert dbg(e._h, kid, out.ctypes.data, cap) == 0
    m = out[:, 1] > 0
    nw = int(m.sum())
    ntt = (ny + stride - 1) // stride
    dur = ((out[:nw, 1].astype(np.int64) - out[:nw, 0].astype(np.int64)) / 100.0)
    t0 = out[:nw, 0].astype(np.int64); start = (t0 - t0.min()) / 100.0
    end = (out[:nw, 1].astype(np.int64) - t0.min()) / 100.0
    nch = nw // ntt
    D = dur[:nch * ntt].reshape(nch, ntt)
    S = start[:nch * ntt].reshape(nch, ntt)
    print("== %s %dx%d %s ic %d: %d waves = %d chunk rows x %d tile columns, R = %.1f rows; span %.1f us, wave mean %.1f max %.1f" % (
        name, nx, ny, dtype, ic, nw, nch, ntt, nx / nch, end.max(), dur.mean(), dur.max()))
    print("   mean duration per chunk row (top to bottom, groups of %d): %s" % (max(1, nch // 16), " ".join("%.0f" % D[k:k + max(1, nch // 16)].mean() for k in range(0, nch, max(1, nch // 16)))))
    print("   mean start   per chunk row                              : %s" % " ".join("%.0f" % S[k:k + max(1, nch // 16)].mean() for k in range(0, nch, max(1, nch // 16))))
    print("   mean duration per tile column: %s" % " ".join("%.0f" % D[:, j].mean() for j in range(ntt)))
    print("   std within a chunk row (mean over rows) %.1f, std of row means %.1f, std of column means %.1f" % (D.std(axis=1).mean(), D.mean(axis=1).std(), D.mean(axis=0).std()))
