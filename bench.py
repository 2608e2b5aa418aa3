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
# k_momentum 6 (F,u,v -> u*,v*,rhs) + 2 x k_jacobi_tb 3 + k_transport 7 (F,u*,v*,p -> F'',u,v: update_uv
# and both FCT sweeps in one pass) = 19 on a full domain; a strip runs the two sweeps as two kernels
# (first sweep with update_uv 7 + second sweep 3, u / v leave for the neighbours in between) = 22
ARRAYS_PER_STEP_FULL = 19
ARRAYS_PER_STEP_STRIP = 22


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
    ap.add_argument("--jacobi-iters", type=int, default=10, help="sweeps per step (reference: 10, 2dvof.py:521)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scaling-reference", action="store_true",
                    help="skip the single-GPU 8192^2 leg (strong_scaling_reference_n1); used for the rocprofv3 "
                         "profiles, whose per-kernel averages must come from one grid size")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the torch.distributed/StripSolver code path even with one rank (self-test)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    ap.add_argument("--dt", type=float, default=0.0,
                    help="time step (default: the reference's 4e-6, 2dvof.py:33, up to 4096^2; 1e-6 at 8192^2, where "
                         "4e-6 exceeds the explicit viscous limit dx^2/(4 nu_g) = 2.5e-6 and the reference algorithm "
                         "-- oracle and GPU alike -- overflows within 10 steps)")
    ap.add_argument("--exchange", default="native", choices=["native", "torch"],
                    help="N > 1: halo exchange by the library's own RCCL communicator (no torch in the process) "
                         "or by torch.distributed P2P")
    ap.add_argument("--attempt-timeout", type=float, default=0.0,
                    help="N > 1: seconds one attempt (native, then torch) may take before the supervising process "
                         "kills it and tries the next carrier (default 300 + 0.01 per step; three times that for torch)")
    ap.add_argument("--no-supervisor", action="store_true", help="N > 1: run in this process, no watchdog / fallback")
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-balance", action="store_true",
                    help="N > 1 (native exchange): keep equal strips instead of re-cutting them by the measured cost of "
                         "each rank's rows (strips.balanced_partition)")
    ap.add_argument("--overlap", type=int, default=4, choices=[0, 1, 2, 3, 4],
                    help="N > 1: 0 = one exchange after the step, 1 = each field as soon as it is final, "
                         "2 = 1 + F's edge bands first, 3 = p, u, v together after the first sweep, 4 = fused transport "
                         "kernel on the edge bands, all four fields in one group under the transport of the other rows "
                         "(vof_step_exchange)")
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
    api = _abi.bind(ctypes.CDLL(so), "ovof_", optional=_abi.GPU_ONLY)
    e = Engine(api, make_desc(api, nx, ny, dtype, "f32"))
    e.set_init_F(ic)
    e.step(1)                      # first touch of every page
    t0 = time.perf_counter()
    e.step(2)
    t1 = (time.perf_counter() - t0) / 2
    n = max(1, min(2000, int(target_s / max(t1, 1e-6))))
    t0 = time.perf_counter()
    e.step(n)
    dt = time.perf_counter() - t0
    e.close()
    out_fast = None
    try:  # also the vectorised build (-O3 -march=native, contraction allowed --
        # NOT bit-identical, timing only), compiled on this host because of -march=native
        import tempfile
        tmp = tempfile.mkdtemp(prefix="vof_oracle_fast_")
        so_fast = os.path.join(tmp, "libvof_oracle_fast.so")
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-shared", "-o", so_fast,
                               os.path.join(ROOT, "oracle", "vof_oracle.c"), "-lm"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        apif = _abi.bind(ctypes.CDLL(so_fast), "ovof_", optional=_abi.GPU_ONLY)
        ef = Engine(apif, make_desc(apif, nx, ny, dtype, "f32"))
        ef.set_init_F(ic)
        ef.step(1)
        nf = max(1, n // 3)
        t0 = time.perf_counter()
        ef.step(nf)
        dtf = time.perf_counter() - t0
        ef.close()
        out_fast = {"value": nx * ny * nf / dtf, "unit": "cell-updates/s", "cores": cores,
                    "build": "gcc -O3 -march=native -fopenmp (not bit-identical)", "steps": nf}
    except Exception:
        pass
    return {"value": nx * ny * n / dt, "unit": "cell-updates/s", "cores": cores, "kind": "port",
            "fast_build": out_fast,
            "sample": "%dx%d %s dam-break, %d steps after 3 warm-up steps, oracle/vof_oracle.c "
                      "(-O2 -ffp-contract=off, OpenMP %d threads), %.1f s" % (nx, ny, dtype, n, cores, dt),
            "ms_per_step": 1e3 * dt / n}


class _StdoutToStderr:
    """Route file descriptor 1 to stderr while RCCL initialises: its C-level banner would otherwise
    land on stdout next to the one JSON line this script must print."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        try:
            ctypes.CDLL(None).fflush(None)   # C stdio buffers of the libraries
        except Exception:
            pass
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def supervise(a, rank):
    """N > 1: run the measurement in a child process per attempt.  This process never touches the
    GPU; it only watches the clock.  If the in-library RCCL path (send/recv groups captured into the
    step graph) should hang or die on a machine it has not been tried on, every rank's supervisor
    times out alike, kills its child and starts the torch.distributed carrier instead, so the run
    still produces its line.  The workers of one attempt find each other through the launcher's pid
    (VOF2D_RDZV_TAG) and the attempt number."""
    import signal
    carriers = ["native", "torch"] if a.exchange == "native" else ["torch"]
    # generous: on a fresh box the first load of the HIP runtime and of RCCL (a 570 MB library) can
    # take a minute; the limit only matters if the attempt hangs
    limit = a.attempt_timeout if a.attempt_timeout > 0 else 300.0 + 0.01 * (a.steps + a.warmup)
    argv = [x for x in sys.argv[1:] if x not in ("--child",)]
    # drop a user-given --exchange (the attempt decides), keep everything else
    cleaned, skip = [], False
    for x in argv:
        if skip:
            skip = False
            continue
        if x == "--exchange":
            skip = True
            continue
        if x.startswith("--exchange="):
            continue
        cleaned.append(x)
    def die_with_parent():
        # the child must not outlive this process (a launcher that tears its workers down would
        # otherwise leave GPU-holding orphans): PR_SET_PDEATHSIG = 1
        ctypes.CDLL(None).prctl(1, int(signal.SIGKILL), 0, 0, 0)

    current = {"p": None}

    def on_term(signum, frame):
        p = current["p"]
        if p is not None and p.poll() is None:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
        sys.exit(128 + signum)
    signal.signal(signal.SIGTERM, on_term)
    signal.signal(signal.SIGINT, on_term)
    for attempt, carrier in enumerate(carriers):
        env = dict(os.environ, VOF2D_RDZV_TAG="%d_%d" % (os.getppid(), attempt))
        cmd = [sys.executable, os.path.abspath(__file__)] + cleaned + ["--child", "--exchange", carrier]
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, start_new_session=True, preexec_fn=die_with_parent)
        current["p"] = p
        try:
            out, _ = p.communicate(timeout=limit if carrier == "native" else 3 * limit)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, signal.SIGKILL)     # exactly the group this process started
            except ProcessLookupError:
                pass
            p.wait()
            print("[bench] rank %d: %s attempt exceeded %.0f s, killed" % (rank, carrier, limit), file=sys.stderr)
            continue
        if p.returncode == 0:
            sys.stdout.write(out.decode())
            sys.stdout.flush()
            return 0
        print("[bench] rank %d: %s attempt exited with code %d" % (rank, carrier, p.returncode), file=sys.stderr)
    return 1


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
    if (world > 1 or a.force_dist) and not a.child and not a.no_supervisor:
        raise SystemExit(supervise(a, rank))
    if os.environ.get("VOF2D_BENCH_TEST_HANG") == a.exchange and a.child:   # self-test of the watchdog
        time.sleep(1e6)
    nx = a.nx or (4096 if world == 1 else 8192)
    ny = a.ny or nx
    esz = 8 if a.dtype == "f64" else 4
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    def stable_dt(n):
        return a.dt if a.dt > 0 else (4e-6 if n <= 4096 else 1e-6)
    dt = stable_dt(max(nx, ny))

    dist_path = world > 1 or a.force_dist
    comm = None
    exchange = "none"
    if not dist_path:
        # single GPU: no torch in the process at all -- ctypes -> C ABI -> HIP
        from vof2d._lib import hip_api
        from vof2d.engine import Engine, make_desc
        api = hip_api()
        eng = Engine(api, make_desc(api, nx, ny, a.dtype, "f32", device=local, jacobi_iters=a.jacobi_iters, dt=dt))
        eng.set_init_F(a.ic)
        eng.step(a.warmup)
        eng.sync()
        t0 = time.perf_counter()
        eng.step(a.steps)
        eng.sync()
        elapsed = time.perf_counter() - t0
        solver = None
    native_ok = False
    if dist_path and a.exchange == "native":
        # one process per GPU, still no torch: rank / world from the launcher's environment, the
        # library's own RCCL communicator for halos, barrier and the max over ranks
        from vof2d.comms import EnvComm
        from vof2d.strips import StripSolver
        try:
            comm = EnvComm(rank, world, local)
            def make_solver(parts=None):
                with _StdoutToStderr():
                    return StripSolver(nx, ny, a.dtype, ic=a.ic, rank=rank, world=world, device=local, parts=parts,
                                       jacobi_iters=a.jacobi_iters, comm=comm, exchange="native" if world > 1 else "auto", dt=dt)
            solver = make_solver()
            parts = solver.parts
            if (world > 1 or os.environ.get("VOF2D_BENCH_TEST_BALANCE")) and not a.no_balance:   # (env: self-test of this block with one rank)
                # Strips do not cost the same: rows of gas take the sweeps' zero shortcuts, the liquid and
                # the interface do not (and GPUs differ a little).  Time each rank's own kernels on the
                # equal strips (no exchange: the halos go stale, the state is thrown away), re-cut the rows
                # so that every rank gets the same share of the cost, and start again from the initial
                # condition.  Results do not depend on the partition.
                import pickle
                from vof2d.strips import balanced_partition
                for _round in range(2):    # the second cut corrects what the piecewise-uniform cost model of the first missed
                    solver.eng.step(3)
                    solver.eng.sync()
                    t0 = time.perf_counter()
                    solver.eng.step(8)
                    solver.eng.sync()
                    cost = (time.perf_counter() - t0) / 8
                    costs = comm.gather_object(cost)
                    blob = None
                    if rank == 0:
                        try:
                            if max(costs) <= (1.015, 1.03)[_round] * sum(costs) / len(costs):
                                blob = pickle.dumps(None)          # balanced within 1.5 % (3 % after one cut: chunk
                                                                   # lengths quantise a strip's cost): keep these strips
                            else:
                                blob = pickle.dumps(balanced_partition(nx, parts, costs, min_rows=solver.halo))
                        except Exception as exc:      # keep every rank on the same partition whatever happens here
                            print("[bench] cost balancing failed (%r): keeping the strips" % (exc,), file=sys.stderr)
                            blob = pickle.dumps(None)
                    new_parts = pickle.loads(comm.broadcast_bytes(blob))
                    solver.barrier()
                    with _StdoutToStderr():
                        solver.close()
                    if new_parts is not None:
                        parts = new_parts
                    solver = make_solver(parts)       # from the initial condition again (the timing steps let the halos go stale)
                    if new_parts is None:
                        break
            native_ok = True
        except Exception as exc:   # e.g. no loadable RCCL: symmetric on all ranks -> the torch carrier
            print("[bench] native RCCL exchange unavailable (%r); falling back to torch.distributed" % (exc,), file=sys.stderr)
    if not dist_path:
        pass
    elif native_ok:
        with _StdoutToStderr():
            eng = solver.eng
            solver.step(a.warmup, overlap=a.overlap)
            eng.sync()
            solver.barrier()
        t0 = time.perf_counter()
        solver.step(a.steps, overlap=a.overlap)
        eng.sync()          # the compute stream has joined the communication stream of every step
        solver.barrier()
        elapsed = time.perf_counter() - t0
        elapsed = comm.allreduce_max(elapsed, eng)
        exchange = solver.exchange_kind if world > 1 else "none"
    else:
        import torch
        import torch.distributed as dist
        from vof2d.strips import StripSolver
        if world == 1:  # --force-dist without a launcher
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local)
        with _StdoutToStderr():
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            solver = StripSolver(nx, ny, a.dtype, ic=a.ic, rank=rank, world=world, device=local,
                                 jacobi_iters=a.jacobi_iters, exchange="torch", dt=dt)
            eng = solver.eng
            solver.step(a.warmup, overlap=bool(a.overlap))
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver.step(a.steps, overlap=bool(a.overlap))
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        exchange = "torch" if world > 1 else "none"

    # Jacobi kernels.
    # (1) the north-star kernel: k_jacobi, one sweep per launch, 3 array passes, HBM-bound.  Timed
    #     live with one HIP event pair on the stream the kernels are launched on, around
    #     --jacobi-sweeps-timed back-to-back launches (vof_time_jacobi).  Agrees with rocprofv3.
    # (2) what the step actually runs: k_jacobi_tb, `tb` sweeps fused per launch.  Its
    #     duration depends on the clock state: sustained back-to-back launches (event pair, same
    #     call) run ~20 % slower than the two launches interleaved in a real step, so the in-step
    #     cost is also derived from wall-clock step times with and without the pressure sweeps.
    comp_rows = min(nx, eng.row_hi - 1) - max(1, eng.row_lo + 1) + 1
    sweep_bytes = 3 * esz * comp_rows * ny          # read p, read rhs, write p' per computed cell
    tb = int(eng.get_param("jacobi_tb"))
    nt = max(2 * tb, a.jacobi_sweeps_timed // (2 * tb) * 2 * tb)
    ms_sweep_tb = eng.time_jacobi(nt)
    eng.set_param("jacobi_tb", 1)
    ms_sweep_1 = eng.time_jacobi(max(2, a.jacobi_sweeps_timed // 2 * 2))
    eng.set_param("jacobi_tb", tb)
    violations = eng.get_counter("courant_violations")
    achieved_1 = sweep_bytes / (ms_sweep_1 * 1e-3) / 1e9
    traffic = load_pmc_traffic(nx, ny, a.dtype) if not dist_path else {}
    fused = {"kernel": "k_jacobi_tb", "sweeps_per_launch": tb, "bound": "hbm (actual traffic: lead-in rows + tile overlap)",
             "us_per_launch_back_to_back": 1e3 * ms_sweep_tb * tb, "us_per_sweep_back_to_back": 1e3 * ms_sweep_tb,
             "hbm_traffic_bytes_per_launch": traffic.get("tb")}
    if not dist_path and a.jacobi_iters > 0 and a.jacobi_iters % tb == 0:
        # in-step cost: (step with sweeps - step without sweeps) / launches, both graph-replayed
        from vof2d.engine import Engine as _E, make_desc as _md
        e0 = _E(api, _md(api, nx, ny, a.dtype, "f32", device=local, jacobi_iters=0, dt=dt))
        e0.set_init_F(a.ic)
        e0.step(a.warmup)
        e0.sync()
        t0 = time.perf_counter()
        e0.step(a.steps)
        e0.sync()
        ms0 = 1e3 * (time.perf_counter() - t0) / a.steps
        e0.close()
        launches = a.jacobi_iters // tb
        us_in_step = (1e3 * elapsed / a.steps - ms0) * 1e3 / launches
        fused.update({"us_per_launch_in_step": us_in_step, "us_per_sweep_in_step": us_in_step / tb,
                      "ms_per_step_without_sweeps": ms0,
                      "algorithmic_GBs_in_step": sweep_bytes * tb / (us_in_step * 1e-6) / 1e9})
    # The N > 1 runs strong-scale 8192^2; give the single-GPU figure for that grid too, so the
    # scaling series has its own N = 1 point (only when the workload was not overridden).
    ref8192 = None
    if not dist_path and not a.nx and rank == 0 and not a.no_scaling_reference:
        try:
            from vof2d.engine import Engine as _E2, make_desc as _md2
            e8 = _E2(api, _md2(api, 8192, 8192, a.dtype, "f32", device=local, jacobi_iters=a.jacobi_iters, dt=stable_dt(8192)))
            e8.set_init_F(a.ic)
            e8.step(3)
            e8.sync()
            t0 = time.perf_counter()
            e8.step(12)
            e8.sync()
            dt8 = time.perf_counter() - t0
            e8.close()
            ref8192 = {"workload": "8192x8192 -ic %d %s dt %g, single strip (the grid bench.py --gpus N > 1 strong-scales)" % (
                a.ic, a.dtype, stable_dt(8192)), "value": 8192 * 8192 * 12 / dt8, "unit": "cell-updates/s", "ms_per_step": 1e3 * dt8 / 12,
                "steps": 12}
        except Exception as exc:   # e.g. not enough free HBM
            ref8192 = {"error": str(exc)}
    prof = eng.profile_steps(14) if not dist_path else {}
    kernels_us = {k: round(v[0], 2) for k, v in prof.items()}

    try:
        one_kernel_transport = (bool(eng.get_param("fuse_transport")) and exchange == "none") or \
                               (exchange == "native" and a.overlap == 4)
    except Exception:
        one_kernel_transport = False
    ARRAYS_PER_STEP = ARRAYS_PER_STEP_FULL if one_kernel_transport else ARRAYS_PER_STEP_STRIP
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
            "config": {"workload": "%dx%d -ic %d %s, dt %g, %d Jacobi sweeps/step, %s" % (
                nx, ny, a.ic, a.dtype, dt, a.jacobi_iters, "single strip" if not dist_path else
                "%d row strips, %d-row deep halo, per-field RCCL send/recv (%s) overlap mode %d" % (
                    world, solver.halo, exchange, a.overlap)),
                "nx": nx, "ny": ny, "dt": dt, "jacobi_iters": a.jacobi_iters,
                "exchange": exchange, "overlap": a.overlap if dist_path else None,
                "rows_per_rank": [hi - lo + 1 for lo, hi in solver.parts] if dist_path else None,
                "exchange_graph": (eng.comm_info()[1] == 1) if exchange == "native" else None,
                "arrays_per_cell_update": ARRAYS_PER_STEP,
                "bytes_per_cell_update_algorithmic": ARRAYS_PER_STEP * esz},
            # The Poisson Jacobi kernel (north star): algorithmic bytes = 3 arrays x sizeof(T) x
            # cells per launch, duration from the HIP-event pair above; `traffic` = HBM
            # bytes per launch from the committed rocprofv3 --pmc passes (profiles/jacobi_pmc.json).
            "roofline": {"bound": "hbm", "kernel": "k_jacobi", "achieved": achieved_1, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved_1 / HBM_PEAK_GBS, "traffic": traffic.get("single"),
                         "us_per_launch": 1e3 * ms_sweep_1, "algorithmic_bytes_per_launch": sweep_bytes,
                         "launches_timed": max(2, a.jacobi_sweeps_timed // 2 * 2)},
            "jacobi_fused": fused,
            "strong_scaling_reference_n1": ref8192,
            "step_hbm_gbs_algorithmic": ARRAYS_PER_STEP * esz * nx * ny * a.steps / elapsed / 1e9,
            "kernels_us_dispatch_start_to_stop": kernels_us,
            "courant_violations": violations,
        }
        if not dist_path and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(nx, ny, a.dtype, a.ic, a.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist_path and native_ok:
        solver.barrier()
        with _StdoutToStderr():
            solver.close()
        comm.cleanup()
    elif dist_path:
        import torch.distributed as dist
        dist.barrier()
        solver.close()
        dist.destroy_process_group()
    else:
        eng.close()


if __name__ == "__main__":
    main()
