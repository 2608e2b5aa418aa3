#!/usr/bin/env python3
"""Cost of the per-step halo exchange around one strip, measured on ONE GPU.

The strip is the one an interior rank of an N-way decomposition of nx x ny owns; both neighbours
are looped back to the rank itself (VOF_COMM_LOOPBACK / self send-recv in torch), so the bytes,
the number of RCCL operations and all host work are those of a real interior rank; only the wire
is missing (device-local copies instead of xGMI).  Reported per step: host enqueue time and wall
time of  compute only | in-library exchange (overlapped / after the step) | torch.distributed
P2P batches (overlapped / after the step).

    python tools/p2p_overhead.py [--n 8 --nx 8192 --ny 8192 --steps 200]
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "taichi-2d-vof_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8); ap.add_argument("--nx", type=int, default=8192)
    ap.add_argument("--ny", type=int, default=8192); ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--dt", type=float, default=1e-6)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--modes", default="compute,native-overlap,native-two,native-after,native-fused,compute-pairs,native-pairs")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--param", action="append", default=[], help="knob=value set on the strip's handle (e.g. tm_rows=52)")
    a = ap.parse_args()
    modes = a.modes.split(",")
    need_torch = any(m.startswith("torch") for m in modes)
    from vof2d import _abi
    from vof2d._lib import hip_api
    from vof2d.engine import Engine, make_desc, comm_unique_id
    from vof2d.strips import partition, stored_rows, _DevArray
    api = hip_api()            # before torch: the process then runs the system HIP / RCCL
    W = _abi.halo_rows(10)
    own = partition(a.nx, a.n)[a.n // 2]
    rows = stored_rows(a.nx, own, W)
    stream_ptr, stream = None, None
    if need_torch:
        import torch, torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        stream = torch.cuda.Stream()
        stream_ptr = stream.cuda_stream
    e = Engine(api, make_desc(api, a.nx, a.ny, a.dtype, "f32", rows=rows, own=own, device=0, dt=a.dt), stream=stream_ptr)
    for kv in a.param:
        e.set_param(kv.split("=")[0], float(kv.split("=")[1]))
    e.set_init_F(1)
    e.step(1)                  # (the first step after set_init_F: the schedule with the reference's intermediate set_BC calls)
    e.comm_init(comm_unique_id(api), 0, 1, loopback=True)
    print("RCCL version code %d, exchange graphs %s" % e.comm_info())
    lo, hi = own[0] - rows[0], own[1] - rows[0]
    if need_torch:
        views = {}
        for f in ("F", "u", "v", "p"):
            base, pitch, col0, nrows = e.field_view(f)
            views[f] = torch.as_tensor(_DevArray(base, (nrows, pitch), "<f8" if a.dtype == "f64" else "<f4"), device="cuda:0")

        def ops_for(f):
            t = views[f]
            return [dist.P2POp(dist.isend, t[lo:lo + W], 0), dist.P2POp(dist.irecv, t[lo - W:lo], 0),
                    dist.P2POp(dist.isend, t[hi - W + 1:hi + 1], 0), dist.P2POp(dist.irecv, t[hi + 1:hi + 1 + W], 0)]
        ops = {f: ops_for(f) for f in views}

        def batch(fields):
            return dist.batch_isend_irecv([o for f in fields for o in ops[f]])

    def run(mode, n):
        if mode == "compute":
            e.step(n)
        elif mode == "native-overlap":
            e.step_exchange(n, 1)
        elif mode == "native-two":
            e.step_exchange(n, 3)
        elif mode == "native-after":
            e.step_exchange(n, 0)
        elif mode == "native-fused":
            e.step_exchange(n, 4)
        elif mode == "native-pairs":          # overlap mode 5: k_jacobi_pair + k_tm, F u* v* rhs p once per step
            e.step_exchange(n, 5)
        elif mode == "compute-pairs":         # ... its kernels without the exchange
            e.step_tm_piece(0)
            for _ in range(n - 1):
                e.step_tm_piece(1)
            e.step_tm_piece(2)
        else:
            with torch.cuda.stream(stream):
                for _ in range(n):
                    if mode == "torch-overlap":
                        e.step_phase(0); w = batch(("p",))
                        e.step_phase(1); w += batch(("u", "v"))
                        e.step_phase(2); w += batch(("F",))
                    else:
                        e.step_phase(0); e.step_phase(1); e.step_phase(2)
                        w = batch(("F", "u", "v", "p"))
                    for x in w:
                        x.wait()

    print("strip %s of %dx%d (own %s), W=%d, %d KiB per field and edge" % (rows, a.nx, a.ny, own, W, W * e.field_view("F")[1] * (8 if a.dtype == "f64" else 4) // 1024))
    for rnd in range(a.rounds):
        for mode in modes:
            run(mode, 10); e.sync()
            t0 = time.perf_counter(); run(mode, a.steps); t_host = time.perf_counter() - t0
            e.sync(); t_all = time.perf_counter() - t0
            print("%-15s host-enqueue %7.1f us/step   wall %7.1f us/step" % (mode, 1e6 * t_host / a.steps, 1e6 * t_all / a.steps), flush=True)
    e.comm_destroy()
    if need_torch:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
