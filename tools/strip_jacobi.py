#!/usr/bin/env python3
"""Jacobi kernels on ONE strip of an N-way row decomposition (default: rank 3 of 8 at 8192^2):
single-sweep kernel vs the fused 5-sweep kernel (both tile widths, several chunk lengths), timed
back to back with one event pair (vof_time_jacobi) a few steps after the start (no tiny values yet)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "taichi-2d-vof_amd"))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8); ap.add_argument("--rank", type=int, default=3)
    ap.add_argument("--nx", type=int, default=8192); ap.add_argument("--ny", type=int, default=8192)
    ap.add_argument("--skip", type=int, default=5)
    a = ap.parse_args()
    from vof2d._lib import hip_api
    from vof2d import _abi
    from vof2d.engine import Engine, make_desc
    from vof2d.strips import partition, stored_rows
    api = hip_api()
    own = partition(a.nx, a.n)[a.rank if a.n > 1 else 0]
    rows = stored_rows(a.nx, own, _abi.halo_rows(10)) if a.n > 1 else None
    dt = 4e-6 if max(a.nx, a.ny) <= 4096 else 1e-6
    kw = dict(rows=rows, own=own) if a.n > 1 else {}
    e = Engine(api, make_desc(api, a.nx, a.ny, "f64", "f32", device=0, dt=dt, **kw))
    e.set_init_F(1); e.step(a.skip); e.sync()
    nrows = (rows[1] - rows[0] + 1) if rows else a.nx
    cells = nrows * a.ny
    def t(label, n=40):
        e.time_jacobi(n)
        us = min(e.time_jacobi(n) for _ in range(3)) * 1e3
        print("%-34s %7.2f us/sweep  %7.1f us/launch-of-5  %6.0f GB/s (24 B rule)" % (label, us, 5 * us, 24 * cells / us / 1e3), flush=True)
    e.set_param("jacobi_tb", 1); t("k_jacobi (1 sweep)")
    e.set_param("jacobi_tb", 5)
    for narrow in (0, 1):
        e.set_param("jacobi_tb_narrow", narrow)
        e.set_param("jacobi_tb_rows", 0); t("tb5 narrow=%d rows=auto" % narrow)
    e.set_param("jacobi_tb_narrow", 0)
    for r in (16, 22, 24, 32, 44, 48, 64, 96):
        e.set_param("jacobi_tb_rows", r); t("tb5 V=2 rows=%d" % r)

if __name__ == "__main__":
    main()
