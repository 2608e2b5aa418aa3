#!/usr/bin/env python3
"""How evenly do the waves of one launch finish?  (diagnostic build: make -C taichi-2d-vof_amd/csrc wavetimes)

    python tools/wave_balance.py [--nx 4096 --ny 4096] [--n 8 --rank 1] [--at 20,200]

For each marching kernel of the fused step: per-wave start/end stamps (s_memrealtime) of one launch ->
launch span, wave durations, and the number of waves in flight over the launch (10 slices)."""
import argparse, ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "taichi-2d-vof_amd"))
KIDS = {"k_momentum": 0, "k_jacobi_tb": 3, "k_transport": 12, "k_fct_x": 5, "k_fct_y": 6}

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=4096); ap.add_argument("--ny", type=int, default=4096)
    ap.add_argument("--n", type=int, default=1); ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--at", default="20,200", help="step numbers at which to sample")
    ap.add_argument("--ic", type=int, default=1)
    ap.add_argument("--extra", type=int, default=2, help="steps run per kernel before its stamps are read (1 or 2: which sweep order comes last)")
    a = ap.parse_args()
    from vof2d import _abi
    from vof2d.engine import Engine, make_desc
    from vof2d.strips import partition, stored_rows
    lib = C.CDLL(os.path.join(ROOT, "taichi-2d-vof_amd", "csrc", "build", "variants", "libvof2d_wavetimes.so"))
    api = _abi.bind(lib, "vof_")
    dbg = lib.vof_debug_wave_times
    dbg.restype = C.c_int; dbg.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32]
    kw = {}
    if a.n > 1:
        own = partition(a.nx, a.n)[a.rank]
        kw = dict(rows=stored_rows(a.nx, own, _abi.halo_rows(10)), own=own)
    dt = 4e-6 if max(a.nx, a.ny) <= 4096 else 1e-6
    e = Engine(api, make_desc(api, a.nx, a.ny, "f64", "f32", device=0, dt=dt, **kw))
    e.set_init_F(a.ic)
    cap = 1 << 16
    done = 0
    for at in [int(x) for x in a.at.split(",")]:
        e.step(at - done); done = at
        print("== %dx%d%s after %d steps" % (a.nx, a.ny, " strip %d/%d" % (a.rank, a.n) if a.n > 1 else "", at))
        for name, kid in KIDS.items():
            h = e._h
            assert dbg(h, kid, None, cap) == 0
            e.step(a.extra); done += a.extra   # the later launch of a kernel overwrites the earlier
            out = np.zeros((cap, 2), np.uint64)
            assert dbg(h, kid, out.ctypes.data, cap) == 0
            m = out[:, 1] > 0
            t0 = out[m, 0].astype(np.int64); t1 = out[m, 1].astype(np.int64)
            if not len(t0):
                print("  %-12s no waves recorded" % name); continue
            # the stamps of the last launch only (an earlier launch of the same kernel, one step
            # before, wrote the same wave ids; if it had more waves its surplus survives)
            nraw = len(t0)
            order = np.argsort(t0); gaps = np.diff(t0[order])
            if len(gaps) and gaps.max() > 15000:   # > 150 us between consecutive wave starts
                cut = t0[order][np.argmax(gaps) + 1]
                sel = t0 >= cut
                t0, t1 = t0[sel], t1[sel]
            if os.environ.get("WB_DEBUG"):
                print("    raw %d kept %d; largest start gaps (us): %s" % (nraw, len(t0), np.sort(gaps)[-4:] / 100.0))
            base = t0.min(); span = (t1.max() - base) / 100.0
            dur = (t1 - t0) / 100.0
            edges = np.linspace(0, t1.max() - base, 11)
            mid = (edges[:-1] + edges[1:]) / 2 + base
            act = [(int(((t0 <= x) & (t1 > x)).sum())) for x in mid]
            busy = dur.sum() / (span * max(act))
            print("  %-12s waves %5d  span %6.1f us  wave us: mean %6.1f p50 %6.1f p90 %6.1f max %6.1f  busy %.2f  in flight: %s" % (
                name, len(t0), span, dur.mean(), np.median(dur), np.percentile(dur, 90), dur.max(), busy,
                " ".join("%d" % x for x in act)), flush=True)

if __name__ == "__main__":
    main()
