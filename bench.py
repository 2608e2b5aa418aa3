#!/usr/bin/env python3
"""bench.py -- cell-updates/s of the fused VOF time step + HBM GB/s of the Jacobi sweep.

    python bench.py --gpus 1 --steps K --warmup W              (4096^2 fp64 dam-break)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
                                                             (8192^2 fp64, N row strips, strong)

A "step" is one pass of the solver part of 2dvof.py's main loop (:506-528): normals + curvature,
momentum predictor, set_BC, rhs, 10 Jacobi sweeps, velocity correction, set_BC, the two FCT
sweeps (+post_process_f), set_BC -- and, for N > 1, the per-step halo exchange.  Inputs are
generated on the device by set_init_F (-ic 1), so they are resident in HBM when timing starts.

Prints ONE JSON line on rank 0 (see DESIGN.md "measurement" for every field).
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "taichi-2d-vof_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md:35
# distinct arrays read or written per cell per step by the fused schedule (DESIGN.md "schedule"):
# k_momentum 6 (F,u,v -> u*,v*,rhs) + 2 x k_jacobi_tb 3 + k_correct 6 (p,F,u*,v* -> u,v) + 2 x k_fct 3
ARRAYS_PER_STEP = 24


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nx", type=int, default=0, help="grid size (default 4096 at 1 GPU, 8192 at N > 1)")
    ap.add_argument("--ny", type=int, default=0)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("-ic", type=int, default=1, choices=[1, 2, 3])
    ap.add_argument("--jacobi-sweeps-timed", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    return ap.parse_args()


def load_pmc_traffic(nx, ny, dtype):
    """HBM bytes per Jacobi launch from the committed rocprofv3 --pmc passes (profiles/jacobi_pmc.json,
    written by tools/summarize_profiles.py) if they were taken at this workload; {} otherwise."""
    path = os.path.join(ROOT, "profiles", "jacobi_pmc.json")
    try:
        rec = json.load(open(path))
        if (rec["nx"], rec["ny"], rec["dtype"]) == (nx, ny, dtype) and isinstance(rec["hbm_bytes_per_launch"], dict):
            return rec["hbm_bytes_per_launch"]
    except Exception:
        pass
    return {}


def usable_cores():
    """Host cores this process may actually use: the affinity mask capped by the cgroup CPU quota
    (the GPU box shows 256 logical CPUs but a 16-CPU quota; 256 OpenMP threads then thrash)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(nx, ny, dtype, ic, target_s):
    """The CPU oracle (scalar-C restatement, OpenMP over i, -O2 -ffp-contract=off = the parity
    build) timed on this host's cores on a bounded sample of the same workload."""
    from vof2d import _abi
    from vof2d.engine import Engine, make_desc
    so = os.path.join(ROOT, "oracle", "_build", "libvof_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    cores = usable_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)   # read by libgomp when the oracle library loads
    api = _abi.bind(ctypes.CDLL(so), "ovof_", optional=("timer_start", "timer_stop", "time_jacobi"))
    e = Engine(api, make_desc(api, nx, ny, dtype, "f32"))
    e.set_init_F(ic)
    t0 = time.perf_counter()
    e.step(1)
    t1 = time.perf_counter() - t0
    n = max(1, min(200, int(target_s / max(t1, 1e-6)) - 1))
    t0 = time.perf_counter()
    e.step(n)
    dt = time.perf_counter() - t0
    e.close()
    return {"value": nx * ny * n / dt, "unit": "cell-updates/s", "cores": cores, "kind": "port",
            "sample": "%dx%d %s dam-break, %d steps after 1 warm-up step, oracle/vof_oracle.c "
                      "(-O2 -ffp-contract=off, OpenMP %d threads), %.1f s" % (nx, ny, dtype, n, cores, dt),
            "ms_per_step": 1e3 * dt / n}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus %d ..." % (a.gpus, a.gpus))
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (a.gpus, world))
    nx = a.nx or (4096 if world == 1 else 8192)
    ny = a.ny or nx
    esz = 8 if a.dtype == "f64" else 4

    if world == 1:
        # single GPU: no torch in the process at all -- ctypes -> C ABI -> HIP
        from vof2d._lib import hip_api
        from vof2d.engine import Engine, make_desc
        api = hip_api()
        eng = Engine(api, make_desc(api, nx, ny, a.dtype, "f32", device=local))
        eng.set_init_F(a.ic)
        eng.step(a.warmup)
        eng.sync()
        t0 = time.perf_counter()
        eng.step(a.steps)
        eng.sync()
        elapsed = time.perf_counter() - t0
        solver = None
    else:
        import torch
        import torch.distributed as dist
        from vof2d.strips import StripSolver
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        solver = StripSolver(nx, ny, a.dtype, ic=a.ic, rank=rank, world=world, device=local)
        eng = solver.eng
        solver.step(a.warmup)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver.step(a.steps)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Jacobi kernels, timed with HIP events on the stream they are launched on (vof_time_jacobi).
    # (1) the kernel the step really uses: k_jacobi_tb, `tb` sweeps fused per launch;
    # (2) the plain one-sweep-per-launch kernel k_jacobi (the HBM-bound form of SURVEY 8d).
    comp_rows = min(nx, eng.row_hi - 1) - max(1, eng.row_lo + 1) + 1
    sweep_bytes = 3 * esz * comp_rows * ny          # read p, read rhs, write p' per computed cell
    tb = int(eng.get_param("jacobi_tb"))
    nt = max(tb * 2, (a.jacobi_sweeps_timed // (2 * tb)) * 2 * tb)
    ms_sweep_tb = eng.time_jacobi(nt)
    eng.set_param("jacobi_tb", 1)
    ms_sweep_1 = eng.time_jacobi(max(2, a.jacobi_sweeps_timed // 2 * 2))
    eng.set_param("jacobi_tb", tb)
    violations = eng.get_counter("courant_violations")
    achieved_tb = sweep_bytes / (ms_sweep_tb * 1e-3) / 1e9      # algorithmic 24 B/cell/sweep rule
    achieved_1 = sweep_bytes / (ms_sweep_1 * 1e-3) / 1e9
    traffic = load_pmc_traffic(nx, ny, a.dtype) if world == 1 else {}

    if rank == 0:
        out = {
            "metric": "cell-updates/sec (whole node), %dx%d %s dam-break" % (nx, ny, "fp64" if esz == 8 else "fp32"),
            "value": nx * ny * a.steps / elapsed,
            "unit": "cell-updates/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": a.dtype,
            "data": "synthetic (set_init_F -ic %d generated on device)" % a.ic,
            "config": {"workload": "%dx%d -ic %d %s, 10 Jacobi sweeps/step, %s" % (
                nx, ny, a.ic, a.dtype, "single strip" if world == 1 else
                "%d row strips, %d-row deep halo, 1 RCCL P2P exchange/step" % (world, solver.halo)),
                "nx": nx, "ny": ny, "jacobi_iters": 10,
                "arrays_per_cell_update": ARRAYS_PER_STEP,
                "bytes_per_cell_update_algorithmic": ARRAYS_PER_STEP * esz},
            # dominant kernel of the step: the fused-sweep Jacobi.  achieved = algorithmic bytes
            # (24 B x cells x sweeps in the launch) / launch time; it can exceed the HBM
            # peak because temporal blocking keeps the intermediate sweeps in registers -- `traffic`
            # is what really crossed HBM per launch (PMC) and hbm_frac_actual the real HBM load.
            "roofline": {"bound": "hbm", "kernel": "k_jacobi_tb<%d sweeps/launch>" % tb,
                         "achieved": achieved_tb, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_tb / HBM_PEAK_GBS,
                         "traffic": traffic.get("tb"),
                         "hbm_frac_actual": (traffic["tb"] / (ms_sweep_tb * tb * 1e-3) / 1e9 / HBM_PEAK_GBS)
                         if traffic.get("tb") else None,
                         "sweeps_per_launch": tb, "us_per_launch": 1e3 * ms_sweep_tb * tb,
                         "us_per_sweep": 1e3 * ms_sweep_tb,
                         "algorithmic_bytes_per_launch": sweep_bytes * tb},
            # the plain single-sweep kernel: genuinely HBM-bound, the north-star ">= 60 % of peak" figure
            "roofline_single_sweep": {"bound": "hbm", "kernel": "k_jacobi", "achieved": achieved_1,
                                      "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_1 / HBM_PEAK_GBS,
                                      "traffic": traffic.get("single"), "us_per_launch": 1e3 * ms_sweep_1,
                                      "algorithmic_bytes_per_launch": sweep_bytes},
            "step_hbm_gbs_algorithmic": ARRAYS_PER_STEP * esz * nx * ny * a.steps / elapsed / 1e9,
            "courant_violations": violations,
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(nx, ny, a.dtype, a.ic, a.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        solver.close()
        dist.destroy_process_group()
    else:
        eng.close()


if __name__ == "__main__":
    main()
