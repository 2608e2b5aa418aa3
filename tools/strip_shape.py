#!/usr/bin/env python3
"""Per-kernel times of ONE strip of an N-way row decomposition, on one GPU.

    python tools/strip_shape.py [--n 8 --rank 3 --nx 8192 --ny 8192 --steps 30]

Builds the handle rank `rank` of `n` would own (owned rows + W halo rows per interior side), runs
the phased step under the in-library profiler (vof_profile_steps) and a wall-clock loop.  Knobs:
--set knob=value (vof_set_param: jacobi_tb, jacobi_tb_rows, momentum_rows, rows_per_wave, ...)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "taichi-2d-vof_amd"))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8); ap.add_argument("--rank", type=int, default=3)
    ap.add_argument("--nx", type=int, default=8192); ap.add_argument("--ny", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=30); ap.add_argument("--dtype", default="f64")
    ap.add_argument("--skip", type=int, default=10, help="steps before the measurement")
    ap.add_argument("--dt", type=float, default=0.0, help="default: 4e-6 up to 4096^2, 1e-6 above (stability, DESIGN.md 4)")
    ap.add_argument("--sweep", default="", help="param=v1,v2,...: wall us/step per value, 3 interleaved rounds")
    ap.add_argument("--kernel", default="", help="with --sweep: report this kernel's profiled us per launch instead of the wall time per step")
    ap.add_argument("--set", action="append", default=[], metavar="KNOB=VALUE", help="vof_set_param before the run")
    ap.add_argument("--lib", default="", help="A/B runs: another build of the library (e.g. csrc/build/variants/libvof2d_base.so)")
    a = ap.parse_args()
    from vof2d._lib import hip_api
    from vof2d import _abi
    from vof2d.engine import Engine, make_desc
    from vof2d.strips import partition, stored_rows
    if a.lib:
        import ctypes
        api = _abi.bind(ctypes.CDLL(os.path.join(ROOT, a.lib)), "vof_")
    else:
        api = hip_api()
    own = partition(a.nx, a.n)[a.rank]
    rows = stored_rows(a.nx, own, _abi.halo_rows(10))
    dt = a.dt if a.dt > 0 else (4e-6 if max(a.nx, a.ny) <= 4096 else 1e-6)
    e = Engine(api, make_desc(api, a.nx, a.ny, a.dtype, "f32", rows=rows, own=own, device=0, dt=dt))
    for kv in a.set:
        e.set_param(kv.split("=")[0], float(kv.split("=")[1]))
    e.set_init_F(1)
    e.step(a.skip); e.sync()
    if a.sweep:
        name, vals = a.sweep.split("=")
        vals = [float(v) for v in vals.split(",")]
        res = {v: [] for v in vals}
        for rnd in range(3):
            for v in vals:
                e.set_param(name, v)
                e.step(6); e.sync()
                if a.kernel:
                    prof = e.profile_steps(8)
                    res[v].append(prof[a.kernel][0] if a.kernel in prof else float("nan"))
                    continue
                t0 = time.perf_counter(); e.step(a.steps); e.sync()
                res[v].append(1e6 * (time.perf_counter() - t0) / a.steps)
        for v in vals:
            print("%s=%-6g  %s us/%s" % (name, v, "  ".join("%.1f" % x for x in res[v]), a.kernel or "step"))
        return
    t0 = time.perf_counter(); e.step(a.steps); e.sync(); dt = (time.perf_counter() - t0) / a.steps
    prof = e.profile_steps(10)
    tot = sum(us * n for us, n in prof.values()) / 10
    print("strip %d/%d rows %s own %s: %.1f us/step wall, kernels %.1f us/step" % (a.rank, a.n, rows, own, 1e6 * dt, tot))
    for k, (us, n) in sorted(prof.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        print("   %-14s %7.1f us x %4.1f /step" % (k, us, n / 10))

if __name__ == "__main__":
    main()
