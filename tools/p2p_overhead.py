#!/usr/bin/env python3
"""Host-side cost of the per-step halo exchange, measured on ONE GPU.

An interior rank's step is 3 x (vof_step_phase + batch_isend_irecv).  With one GPU there is no
neighbour, so the strip posts the same number of send/recv descriptors *to itself* (RCCL supports
self send/recv inside a group): the bytes moved are the real halo bytes, the peer is wrong, the CPU
work (torch P2POp batching, RCCL group launch, stream events) is the same.  Reports wall time per
step of (a) compute only, (b) compute + the three batches, (c) one batch after the step.

    python tools/p2p_overhead.py [--nx 1056 --ny 8192 --steps 200]
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "taichi-2d-vof_amd"))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=1056)   # rows an interior rank of 8 stores at 8192^2
    ap.add_argument("--ny", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=200)
    a = ap.parse_args()
    import torch, torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from vof2d.strips import StripSolver, EXCHANGED
    s = StripSolver(a.nx, a.ny, "f64", ic=1, rank=0, world=1, device=0, dist=dist)
    W = s.halo
    def ops_for(f):
        t, _ = s._views[f]
        lo, hi = 1 + W, a.nx - W
        o = []
        for (src, dst) in (((lo, lo + W), (lo - W, lo)), ((hi - W, hi), (hi, hi + W))):
            o.append(dist.P2POp(dist.isend, t[src[0]:src[1]], 0))
            o.append(dist.P2POp(dist.irecv, t[dst[0]:dst[1]], 0))
        return o
    ops = {f: ops_for(f) for f in EXCHANGED}
    def batch(fields):
        l = []
        for f in fields: l += ops[f]
        return dist.batch_isend_irecv(l)

    def run(mode, n):
        with torch.cuda.stream(s.stream):
            for _ in range(n):
                if mode == "compute":
                    s.eng.step_phase(0); s.eng.step_phase(1); s.eng.step_phase(2)
                elif mode == "overlap3":
                    s.eng.step_phase(0); w = batch(("p",))
                    s.eng.step_phase(1); w += batch(("u", "v"))
                    s.eng.step_phase(2); w += batch(("F",))
                    for x in w: x.wait()
                elif mode == "single":
                    s.eng.step_phase(0); s.eng.step_phase(1); s.eng.step_phase(2)
                    for x in batch(EXCHANGED): x.wait()
                elif mode == "host_only":     # the batches alone, no kernels
                    w = batch(("p",)); w += batch(("u", "v")); w += batch(("F",))
                    for x in w: x.wait()
    out = {}
    for mode in ("compute", "overlap3", "single", "host_only"):
        try:
            run(mode, 10); s.sync(); torch.cuda.synchronize()
            t0 = time.perf_counter(); run(mode, a.steps); t_host = time.perf_counter() - t0
            s.sync(); torch.cuda.synchronize(); t_all = time.perf_counter() - t0
            out[mode] = (1e6 * t_host / a.steps, 1e6 * t_all / a.steps)
            print("%-10s host-enqueue %7.1f us/step   wall %7.1f us/step" % (mode, *out[mode]), flush=True)
        except Exception as e:
            print(mode, "FAILED:", repr(e)[:300], flush=True)
    dist.destroy_process_group()

if __name__ == "__main__":
    main()
