#!/usr/bin/env python3
"""bench.py -- cell-updates/s of the fused VOF time step + HBM GB/s of the Jacobi sweep.

    python bench.py --gpus 1 --steps K --warmup W              (4096^2 fp64 dam-break)
    python bench.py --gpus N --steps K --warmup W              (8192^2 fp64, N row strips, strong;
                                                                starts its own N workers)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...      (same, launcher-started)

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
# algorithmic array passes per launch of the kernels of the fused step (same counting rule)
KERNEL_PASSES = {"k_momentum": 6, "k_jacobi_tb": 3, "k_transport": 7, "k_jacobi": 3, "k_fct_x": 7, "k_fct_y": 7,
                 # k_tm (k_transport + the next step's k_momentum): F, u*, v*, p -> F'', u*', v*', rhs'; k_jacobi_pair: p, rhs -> p after TEN sweeps
                 # k_tm_uv: the last k_tm of a batch, which also stores u and v (the batches chain: no plain k_momentum / k_transport is left)
                 "k_tm": 8, "k_tm_uv": 10, "k_jacobi_pair": 3}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (default 200: 0.12 s at 4096^2)")
    ap.add_argument("--warmup", type=int, default=20,
                    help="untimed steps before them.  The first step after set_init_F runs the eager two-kernel schedule; "
                         "the library uploads its step graphs when it builds them, so no first-launch cost falls into "
                         "the timed region whatever W is")
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
    ap.add_argument("--no-extras", action="store_true",
                    help="N = 1: skip the sustained 1000-step record and the 1024^2 residual-terminated solve "
                         "(the rocprofv3 profiles: one workload per trace)")
    ap.add_argument("--sustained-steps", type=int, default=1000)
    ap.add_argument("--profile-steps", type=int, default=350,
                    help="N = 1: steps of the in-situ kernel profile behind `roofline` and `step_kernels` (a fresh run, steps "
                         "11 .. 10 + this many: with the default it covers the start of the tiny-value front, steps ~65-600)")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the torch.distributed/StripSolver code path even with one rank (self-test)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline's main sample")
    ap.add_argument("--dt", type=float, default=0.0,
                    help="time step (default: the reference's 4e-6, 2dvof.py:33, up to 4096^2; 1e-6 at 8192^2, where "
                         "4e-6 exceeds the explicit viscous limit dx^2/(4 nu_g) = 2.5e-6 and the reference algorithm "
                         "-- oracle and GPU alike -- overflows within 10 steps)")
    ap.add_argument("--exchange", default="native", choices=["native", "torch"],
                    help="N > 1: halo exchange by the library's own RCCL communicator (no torch in the process) "
                         "or by torch.distributed P2P")
    ap.add_argument("--attempt-timeout", type=float, default=0.0,
                    help="N > 1: seconds one attempt (native, then torch) may take before the supervising process "
                         "kills it and tries the next carrier (default 120 + 0.05 per step; three times that for torch)")
    ap.add_argument("--no-supervisor", action="store_true", help="N > 1: run in this process, no watchdog / fallback")
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--fast-leg", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dry-run", action="store_true",
                    help="N > 1: start the workers, let them find each other through the rendezvous directory and "
                         "report -- no GPU work (self-test of the launcher)")
    ap.add_argument("--no-balance", action="store_true",
                    help="N > 1 (native exchange): keep equal strips instead of re-cutting them by the measured cost of "
                         "each rank's rows (strips.balanced_partition)")
    ap.add_argument("--same-device", action="store_true",
                    help="N > 1 REHEARSAL on one GPU: every rank opens device 0.  RCCL refuses to form a communicator of two ranks "
                         "on one device, so the native attempt ends at vof_comm_init (after the rendezvous) and the halos travel by "
                         "the second carrier over gloo, staged through host memory (StripSolver.stage_host): the launcher, the "
                         "supervisor, the rendezvous, the cost re-cut and the assembly of the N > 1 line run on real HIP handles.  "
                         "The rate it prints is of N processes sharing one GPU -- a plumbing check, not a scaling figure")
    ap.add_argument("--digest", action="store_true",
                    help="N > 1: gather F, u, v, p on rank 0 after the timed steps and put their SHA-256 (and the step count) into the "
                         "line (`fields_sha256`), for comparison with a single-domain run of the same number of steps")
    ap.add_argument("--overlap", type=int, default=5, choices=[0, 1, 3, 4, 5],
                    help="N > 1: 0 = one exchange after the step, 1 = each field as soon as it is final, "
                         "3 = p, u, v together after the first sweep, 4 = fused transport "
                         "kernel on the edge bands, all four fields in one group under the transport of the other rows, "
                         "5 (default, both precisions) = the pair kernels of the single GPU -- k_jacobi_pair and k_tm --, F, u*, "
                         "v*, rhs, p exchanged once per step (vof_step_exchange)")
    return ap.parse_args()


def load_pmc_traffic(nx, ny, dtype, path=None):
    """HBM bytes per Jacobi launch from the committed rocprofv3 --pmc passes (profiles/jacobi_pmc.json,
    written by tools/summarize_profiles.py) -- only if they were taken at this workload AND on the
    kernel sources this library was built from (the file records their hash); otherwise nothing is
    quoted and the reason is reported as `traffic_note`.  Returns (bytes per launch by kernel, note)."""
    path = path or os.path.join(ROOT, "profiles", "jacobi_pmc.json")
    try:
        rec = json.load(open(path))
    except Exception:
        return {}, "no committed PMC profile"
    if (rec.get("nx"), rec.get("ny"), rec.get("dtype")) != (nx, ny, dtype) or not isinstance(rec.get("hbm_bytes_per_launch"), dict):
        return {}, "the committed PMC profile (%s) is of another workload" % rec.get("tag")
    from vof2d._lib import kernel_source_hash
    if rec.get("kernel_source_sha256") != kernel_source_hash():
        return {}, "the committed PMC profile (%s) was taken on other kernel sources: re-run tools/summarize_profiles.py" % rec.get("tag")
    return rec["hbm_bytes_per_launch"], "rocprofv3 --pmc passes of profile %s (same kernel sources, same workload), not this run" % rec.get("tag")


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


# --------------------------------------------------------------------------------------------------
# CPU baseline: the oracle's C restatement on this host's cores
def _omp_threads(n):
    """Thread count of the already loaded libgomp (the oracle reads OMP_NUM_THREADS only at load)."""
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except Exception:
        pass


def _time_cpu(api, nx, ny, dtype, ic, threads, target_s, max_steps=2000):
    """(cell-updates/s, steps, seconds, Jacobi GB/s by the 24 B rule) of one oracle build."""
    from vof2d.engine import Engine, make_desc
    _omp_threads(threads)
    esz = 8 if dtype == "f64" else 4
    e = Engine(api, make_desc(api, nx, ny, dtype, "f32"))
    e.set_init_F(ic)
    e.step(1)                      # first touch of every page
    t0 = time.perf_counter()
    e.step(1)
    t1 = time.perf_counter() - t0
    n = max(1, min(max_steps, int(target_s / max(t1, 1e-6))))
    t0 = time.perf_counter()
    e.step(n)
    dt = time.perf_counter() - t0
    # the Poisson sweep alone (:236-266 as written: rhs recomputed + copy-back), same 24 B rule as the GPU's
    e.cal_nu_rho(); e.get_normal_young(); e.advect_upwind(); e.set_BC()
    e.solve_p_jacobi(1)
    ns = max(2, min(200, int(0.15 * target_s / max(t1 / 12.0, 1e-6))))
    t0 = time.perf_counter()
    e.solve_p_jacobi(ns)
    ts = (time.perf_counter() - t0) / ns
    e.close()
    return nx * ny * n / dt, n, dt, 3 * esz * nx * ny / ts / 1e9


def cpu_baseline(nx, ny, dtype, ic, target_s):
    """The CPU oracle (scalar-C restatement, OpenMP over i) timed on this host's cores on a bounded
    sample of the same workload: the parity build (-O2 -ffp-contract=off) on all usable cores and on
    one thread, and the vectorised build (-O3 -march=native, not bit-identical) on all cores."""
    from vof2d import _abi
    so = os.path.join(ROOT, "oracle", "_build", "libvof_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    cores = usable_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)   # read by libgomp when the oracle library loads
    api = _abi.bind(ctypes.CDLL(so), "ovof_", optional=_abi.GPU_ONLY)
    val, n, dt, jac = _time_cpu(api, nx, ny, dtype, ic, cores, target_s)
    one = None
    try:
        v1, n1, dt1, jac1 = _time_cpu(api, nx, ny, dtype, ic, 1, 0.3 * target_s, max_steps=200)
        one = {"value": v1, "unit": "cell-updates/s", "cores": 1, "steps": n1, "seconds": dt1, "jacobi_GBs_24B_rule": jac1}
    except Exception:
        pass
    out_fast = None
    try:  # also the vectorised build (contraction allowed -- NOT bit-identical,
        # timing only), compiled on this host because of -march=native
        import tempfile
        tmp = tempfile.mkdtemp(prefix="vof_oracle_fast_")
        so_fast = os.path.join(tmp, "libvof_oracle_fast.so")
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"),
                               "-o", so_fast, os.path.join(ROOT, "oracle", "vof_oracle.c"), "-lm"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        apif = _abi.bind(ctypes.CDLL(so_fast), "ovof_", optional=_abi.GPU_ONLY)
        vf, nf, dtf, jacf = _time_cpu(apif, nx, ny, dtype, ic, cores, 0.35 * target_s)
        out_fast = {"value": vf, "unit": "cell-updates/s", "cores": cores, "jacobi_GBs_24B_rule": jacf,
                    "build": "gcc -O3 -march=native -fopenmp (not bit-identical)", "steps": nf, "seconds": dtf}
    except Exception:
        pass
    _omp_threads(cores)
    return {"value": val, "unit": "cell-updates/s", "cores": cores, "kind": "port",
            "jacobi_GBs_24B_rule": jac, "threads_1": one, "fast_build": out_fast,
            "sample": "%dx%d %s dam-break, %d steps after 2 warm-up steps, oracle/vof_oracle.c "
                      "(-O2 -ffp-contract=off, OpenMP %d threads), %.1f s; then the same build on 1 thread and the "
                      "-O3 -march=native build on %d threads, each on a shorter sample; jacobi_GBs_24B_rule = 3 arrays x "
                      "sizeof(T) x cells / time of one solve_p_jacobi() call as written in 2dvof.py:236-266" % (
                          nx, ny, dtype, n, cores, dt, cores),
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


from vof2d.launch import die_with_parent as _die_with_parent  # noqa: E402


# --------------------------------------------------------------------------------------------------
def launch_workers(a):
    """`python bench.py --gpus N` with no launcher around it: this process becomes the launcher
    (vof2d/launch.py): it never touches the GPU, starts N fresh workers with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set as torch.distributed.run would, relays rank 0's line."""
    from vof2d.launch import spawn_ranks
    return spawn_ranks(__file__, sys.argv[1:], a.gpus)


def supervise(a, rank):
    """N > 1: run the measurement in a child process per attempt.  This process never touches the
    GPU; it only watches the clock.  If the in-library RCCL path (send/recv groups captured into the
    step graph) should hang or die on a machine it has not been tried on, every rank's supervisor
    times out alike, kills its child and starts the torch.distributed carrier instead, so the run
    still produces its line.  The workers of one attempt find each other through the launcher's pid
    (VOF2D_RDZV_TAG) and the attempt number."""
    import signal
    carriers = ["native", "torch"] if a.exchange == "native" else ["torch"]
    # the limit only matters if the attempt hangs.  Fixed part: on a fresh box the first load of the HIP runtime and of
    # RCCL (a 570 MB library) and ncclCommInitRank can take a minute; proportional part: 50 ms per step is ~100 x what
    # a strip step takes (0.35 ms), and covers the cost probe, the re-cut and the same-grid single-GPU reference leg
    limit = a.attempt_timeout if a.attempt_timeout > 0 else 120.0 + 0.05 * (a.steps + a.warmup)
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
    base_tag = os.environ.get("VOF2D_RDZV_TAG") or str(os.getppid())
    for attempt, carrier in enumerate(carriers):
        env = dict(os.environ, VOF2D_RDZV_TAG="%s_%d" % (base_tag, attempt))
        cmd = [sys.executable, os.path.abspath(__file__)] + cleaned + ["--child", "--exchange", carrier]
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, start_new_session=True, preexec_fn=_die_with_parent)
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


# --------------------------------------------------------------------------------------------------
# single-GPU extras
HANDLE_WARM_STEPS = 19   # 1 eager step + a batch of 16 + a batch of 2: the step graphs of the handle's batch form captured and replayed once


def warm_handle(e, ic, nx, ny, dtype):
    """A handle captures its step graphs on its first steady-state steps (and, on a large fp64 grid, looks at F once to
    choose its batch form: a rule on the state, vof_step) -- once, 20-40 ms all told.  A bench of a few dozen steps is
    not the place to charge that to, so the handle is warmed and then put back to the initial state: all-zero F, u, v,
    p as a new handle holds them, set_init_F (which, like 2dvof.py:141-147, only writes the liquid cells of the dam),
    istep = 0.  tests/test_parity_gpu.py checks that a handle treated like this repeats a new handle's run value for
    value.  Returns (wall time of the warm steps in ms, Courant violations counted during them -- the device counter is
    not reset with the state)."""
    import numpy as np
    e.sync()
    t0 = time.perf_counter()
    e.step(HANDLE_WARM_STEPS)       # (one eager step and an even number of fused ones: the F / twin pair is back in place)
    e.sync()
    ms = 1e3 * (time.perf_counter() - t0)
    zeros = np.zeros((nx + 2, ny + 2), dtype=np.float64 if dtype == "f64" else np.float32)
    for f in ("F", "u", "v", "p"):
        e.set(f, zeros)
    del zeros
    e.set_init_F(ic)
    e.istep = 0
    e.sync()
    return ms, e.get_counter("courant_violations")


def _batch_form(form):
    halves, choice, tm_steps = form
    if choice == 1 or (choice < 0 and tm_steps > 0):
        return "k_tm batch graphs (k_transport + the next step's k_momentum as one kernel)"
    return "chain batch graphs (every kernel as launches on row blocks)" if halves else "one-chain batch graphs"


def timed_steps(eng, warmup, steps):
    eng.step(warmup)
    eng.sync()
    t0 = time.perf_counter()
    eng.step(steps)
    eng.sync()
    return time.perf_counter() - t0


def sustained_record(api, nx, ny, dtype, ic, local, jacobi_iters, dt, nsteps, block=100):
    """ms/step over steps 1..nsteps of a fresh run in blocks of `block` steps (one device sync per
    block): the default timing window (a few dozen steps right after the start) does not see the
    regime in which the decaying front of the pressure iteration crosses the grid."""
    from vof2d.engine import Engine, make_desc
    e = Engine(api, make_desc(api, nx, ny, dtype, "f32", device=local, jacobi_iters=jacobi_iters, dt=dt))
    e.set_init_F(ic)
    warm_ms, warm_viol = warm_handle(e, ic, nx, ny, dtype)
    blocks = []
    t_all = time.perf_counter()
    for _ in range(max(1, nsteps // block)):
        t0 = time.perf_counter()
        e.step(block)
        e.sync()
        blocks.append(1e3 * (time.perf_counter() - t0) / block)
    total = time.perf_counter() - t_all
    n = block * len(blocks)
    viol = e.get_counter("courant_violations") - warm_viol
    e_form = (e.get_param("overlap_halves"), e.get_counter("tm_choice"), e.get_counter("tm_steps"))
    e.close()
    return {"steps": n, "block": block, "ms_per_step_blocks": [round(b, 4) for b in blocks],
            "ms_per_step": 1e3 * total / n, "ms_per_step_worst_block": max(blocks),
            "value": nx * ny * n / total, "unit": "cell-updates/s", "courant_violations": viol,
            "handle_warm_steps_ms": round(warm_ms, 2), "batch_form": _batch_form(e_form),
            "note": "steps 1..%d from the initial state on a warm handle (%d steps, then F = u = v = p = 0, set_init_F, "
                    "istep = 0: its step graphs are captured), wall clock incl. one sync per block" % (n, HANDLE_WARM_STEPS)}


def residual_solve_1024(api, local, dtype="f64", tol=1e-6, cap=3000000, every=5000):
    """BASELINE configs[1]: 1024^2 dam-break, the first pressure solve of the run (p = 0 start) iterated
    until max|p_new - p| / max|p_new| <= 1e-6 (vof_solve_p, relative criterion of SURVEY 8f-1) instead
    of the reference's fixed 10 sweeps.  Five sweeps per launch; the norms are reduced in the last
    launch before each check."""
    from vof2d.engine import Engine, make_desc
    n = 1024
    e = Engine(api, make_desc(api, n, n, dtype, "f32", device=local))
    e.set_init_F(1)
    e.cal_nu_rho(); e.get_normal_young(); e.advect_upwind(); e.set_BC()    # :513-518 of step 1
    e.solve_p_jacobi(10)                                                    # (warm the kernels; 10 sweeps of the solve)
    e.sync()
    t0 = time.perf_counter()
    it, res = e.solve_p(tol, cap, every, "rel")
    ms = 1e3 * (time.perf_counter() - t0)
    e.close()
    esz = 8 if dtype == "f64" else 4
    return {"workload": "1024x1024 -ic 1 %s, pressure solve of step 1 from p = 0" % dtype,
            "criterion": "max|p_new-p| / max(max|p_new|, 1e-300) <= %g, checked every %d sweeps" % (tol, every),
            "iterations": it + 10, "residual": res, "converged": bool(res <= tol), "ms": ms,
            "sweeps_per_s": it / (ms * 1e-3), "us_per_sweep": 1e3 * ms / max(it, 1),
            "algorithmic_GBs_24B_rule": 3 * esz * n * n * it / (ms * 1e-3) / 1e9,
            "note": "1024^2 (3 x 8.4 MB) sits in L2 / MALL: cache bandwidth, not HBM; the relative residual of this "
                    "pure-Neumann iteration decays like 1/k once the null-space constant dominates the update (1e5..1e6 sweeps for 1e-6 on any grid)"}


def single_gpu_reference(api, n, dtype, ic, local, jacobi_iters, dt, steps=12):
    from vof2d.engine import Engine, make_desc
    e = Engine(api, make_desc(api, n, n, dtype, "f32", device=local, jacobi_iters=jacobi_iters, dt=dt))
    e.set_init_F(ic)
    warm_handle(e, ic, n, n, dtype)
    steps = max(steps, 24)
    el = timed_steps(e, 3, steps)
    form = _batch_form((e.get_param("overlap_halves"), e.get_counter("tm_choice"), e.get_counter("tm_steps")))
    e.close()
    return {"workload": "%dx%d -ic %d %s dt %g, single strip (the grid bench.py --gpus N > 1 strong-scales)" % (n, n, ic, dtype, dt),
            "value": n * n * steps / el, "unit": "cell-updates/s", "ms_per_step": 1e3 * el / steps, "steps": steps, "batch_form": form}


def worst_case_record(api, nx, ny, dtype, local, jacobi_iters, dt, steps):
    """The same grid holding the flow the step is slowest on: the rising bubble (-ic 2, 2 % gas).  The headline workload
    (dam-break: 5/6 of the cells are gas) leans on the exact gas-row shortcuts of the transport; a mostly-liquid grid takes none
    of them, so the driver-visible line shows the spread."""
    from vof2d.engine import Engine, make_desc
    e = Engine(api, make_desc(api, nx, ny, dtype, "f32", device=local, jacobi_iters=jacobi_iters, dt=dt))
    e.set_init_F(2)
    warm_handle(e, 2, nx, ny, dtype)
    steps = max(steps, 40)
    el = timed_steps(e, 5, steps)
    form = _batch_form((e.get_param("overlap_halves"), e.get_counter("tm_choice"), e.get_counter("tm_steps")))
    share = e.get_param("gas_share")
    e.close()
    return {"workload": "%dx%d -ic 2 (rising bubble) %s, dt %g" % (nx, ny, dtype, dt), "ms_per_step": 1e3 * el / steps, "steps": steps,
            "value": nx * ny * steps / el, "unit": "cell-updates/s", "gas_share": share, "step_schedule": form}


def fast_leg(a):
    """Child process of the N = 1 run: the same workload on the FMA-contracted build
    (libvof2d_hip_fast.so), and what contraction does to the results: F after 1000
    steps of BASELINE configs[0] (128^2 dam-break fp64) against the committed fixture
    tests/golden/dam128_f64.npz, which the parity build reproduces exactly (north-star bar: 1e-5)."""
    import numpy as np
    from vof2d._lib import hip_api
    from vof2d.engine import Engine, make_desc
    api = hip_api(fast=True)
    nx = a.nx or 4096
    ny = a.ny or nx
    e = Engine(api, make_desc(api, nx, ny, a.dtype, "f32", device=int(os.environ.get("LOCAL_RANK", "0")),
                              jacobi_iters=a.jacobi_iters, dt=a.dt if a.dt > 0 else 4e-6))
    e.set_init_F(a.ic)
    warm_handle(e, a.ic, nx, ny, a.dtype)
    el = timed_steps(e, a.warmup, a.steps)
    e.close()
    out = {"build": "hipcc -ffp-contract=fast (FMA contraction; not bit-identical to the reference's operation order)",
           "value": nx * ny * a.steps / el, "unit": "cell-updates/s", "ms_per_step": 1e3 * el / a.steps}
    try:
        z = np.load(os.path.join(ROOT, "tests", "golden", "dam128_f64.npz"))
        g = Engine(api, make_desc(api, 128, 128, "f64", "f32", device=int(os.environ.get("LOCAL_RANK", "0"))))
        g.set_init_F(1)
        g.step(100)
        d100 = float(np.max(np.abs(g.get("F") - z["F_100"])))
        g.step(900)
        d1000 = float(np.max(np.abs(g.get("F") - z["F_1000"])))
        g.close()
        out.update({"F_Linf_vs_parity_build_128x128_step_100": d100, "F_Linf_vs_parity_build_128x128_step_1000": d1000,
                    "meets_1e-5_bar_at_step_1000": bool(d1000 <= 1e-5)})
    except Exception as exc:
        out["difference_error"] = str(exc)
    print(json.dumps(out), flush=True)


def run_fast_leg(a):
    cmd = [sys.executable, os.path.abspath(__file__), "--fast-leg", "--steps", str(a.steps), "--warmup", str(a.warmup),
           "--dtype", a.dtype, "-ic", str(a.ic), "--jacobi-iters", str(a.jacobi_iters)]
    if a.nx:
        cmd += ["--nx", str(a.nx), "--ny", str(a.ny or a.nx)]
    if a.dt > 0:
        cmd += ["--dt", str(a.dt)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        return {"error": (r.stderr or r.stdout)[-300:]}
    return json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])


def roofline_record(step_kernels, traffic, traffic_note, nprof, prof_note, sweep_bytes, fused, tb, single):
    """The `roofline` object of the line: k_jacobi_tb from the in-situ profile; if the profile is missing (N > 1, or the
    second engine did not fit) the back-to-back figure of vof_time_jacobi stands in and says so."""
    k = step_kernels.get("k_jacobi_tb")
    if k:
        us, src = k["us_per_launch_dispatch"], ("in-situ profile (vof_profile_steps: the four launches of a step on the whole grid, one at a time, "
                                                "HIP events around each), %d steps from step 11 of a fresh run" % nprof)
    else:
        us, src = fused["us_per_launch_back_to_back"], "back-to-back launches (vof_time_jacobi): %s" % (prof_note or "no in-situ profile on this path")
    achieved = sweep_bytes / (us * 1e-6) / 1e9
    low = None
    if step_kernels:
        n = min(step_kernels, key=lambda x: step_kernels[x]["frac_of_peak"])
        low = dict(step_kernels[n], kernel=n, peak=HBM_PEAK_GBS, unit="GB/s", achieved=step_kernels[n]["frac_of_peak"] * HBM_PEAK_GBS)
    return {"bound": "hbm", "kernel": "k_jacobi_tb", "sweeps_per_launch": tb, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic.get("tb"), "traffic_note": traffic_note,
            "us_per_launch": us, "algorithmic_bytes_per_launch": sweep_bytes, "duration_source": src,
            "in_step_lowest": low, "north_star_single_sweep": single}


def roofline_of_the_kept_form(run_kernels, classic, traffic_tm, note_tm, nprof, esz, cells):
    """`roofline` when the handle keeps the k_tm form (large fp64 grids): the step is k_jacobi_pair
    (two five-sweep launches as one) and k_tm (k_transport + the next step's k_momentum as one), and k_tm is the
    dominant kernel -- about two thirds of the step.  Algorithmic bytes by SURVEY 8d's rule (every distinct array a
    kernel reads or writes, once): 8 passes for k_tm, 3 for k_jacobi_pair's ten sweeps.  The rule counts what a
    kernel has to move, so a kernel that fuses more has fewer bytes to show for its time: next to `frac` stand the
    bytes of the kernels it replaces; the whole step on its own kernel list is `step_frac_of_peak_algorithmic`
    (config.step_frac_of_peak: own list, the four-kernel schedule's 19 passes, counter traffic, side by side).
    `one_kernel_at_a_time` is the record of the four classic kernels (k_jacobi_tb: the previous rounds' `roofline`)."""
    tm = run_kernels["k_tm"]
    us = tm["us_per_launch_dispatch"]
    per_step = sum(v["us_per_launch_dispatch"] * v["launches_per_step"] for v in run_kernels.values())
    out = {"bound": "hbm", "kernel": "k_tm", "achieved": tm["frac_of_peak"] * HBM_PEAK_GBS, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": tm["frac_of_peak"], "traffic": traffic_tm.get("tm"), "traffic_note": note_tm, "us_per_launch": us,
           "algorithmic_passes": 8, "algorithmic_bytes_per_launch": tm["algorithmic_bytes_per_launch"],
           "share_of_the_step_kernel_time": us * tm["launches_per_step"] / per_step if per_step else None,
           "frac_on_the_bytes_of_the_two_kernels_it_replaces": (7 + 6) * esz * cells / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
           "duration_source": "in-situ profile (vof_profile_steps) of the launch sequence the handle's batch graphs replay, every launch "
                              "between its own HIP event pair, %d steps from step 11 of a run on the warmed handle" % nprof,
           "step_kernels_as_run": run_kernels, "one_kernel_at_a_time": classic,
           "north_star_single_sweep": classic.get("north_star_single_sweep")}
    jp = run_kernels.get("k_jacobi_pair")
    if jp:
        out["jacobi_kernel_of_the_step"] = dict(jp, kernel="k_jacobi_pair", sweeps_per_launch=10, traffic=traffic_tm.get("pair"),
                                                sweeps_GBs_by_the_24B_rule=10 * 3 * esz * cells / (jp["us_per_launch_dispatch"] * 1e-6) / 1e9,
                                                frac_on_the_bytes_of_the_two_launches_it_replaces=2 * jp["frac_of_peak"])
    return out


def balance_strips(a, comm, solver, make_solver, rank, world, nx):
    """Strips do not cost the same: rows of gas take the sweeps' zero shortcuts, the liquid and the interface do not (and
    GPUs differ a little).  Time each rank's strip on the equal partition, re-cut the rows so that every rank gets the
    same share of the cost, and start again from the initial condition.  Results do not depend on the partition.
    The probe (StripSolver.probe_cost) times the rank's own kernels -- in mode 5 the pair kernels of a middle step, which weigh
    the liquid differently from the four-kernel step -- on VALID data (the kernels are data dependent: zero shortcuts,
    division tiers -- stale halos would feed them garbage rows): one kernel-only step, timed on the device, then a full
    halo exchange before the next one (a deep halo covers exactly one step).  Waiting for neighbours is not in the figure,
    or every rank would read the slowest rank's time.  A failure on one rank is agreed on before anybody changes partition.
    Either carrier (comm: an EnvComm beside the library's own RCCL communicator, or a TorchComm).  Returns the solver to
    time -- a new one, from the initial condition again."""
    if not ((world > 1 or os.environ.get("VOF2D_BENCH_TEST_BALANCE")) and not a.no_balance):   # (env: self-test of this block with one rank)
        return solver
    from vof2d.strips import balanced_partition
    parts = solver.parts
    for _round in range(2):    # the second cut corrects what the piecewise-uniform cost model of the first missed
        cost, failed = 0.0, 0.0
        try:
            with _StdoutToStderr():
                cost = solver.probe_cost(10, 2, a.overlap)    # ms of this rank's stream per step of the kernels the timed run will launch
        except Exception as exc:
            print("[bench] rank %d: cost probe failed (%r)" % (rank, exc), file=sys.stderr)
            failed = 1.0
        failed = comm.allreduce_max(failed, solver.eng)     # every rank leaves the block together
        new_parts = None
        if not failed:
            costs = comm.gather_object(cost)
            decision = None
            if rank == 0:
                try:
                    if max(costs) > (1.015, 1.03)[_round] * sum(costs) / len(costs):
                        decision = balanced_partition(nx, parts, costs, min_rows=solver.halo)
                    # else: balanced within 1.5 % (3 % after one cut: chunk lengths quantise a strip's cost)
                except Exception as exc:      # keep every rank on the same partition whatever happens here
                    print("[bench] cost balancing failed (%r): keeping the strips" % (exc,), file=sys.stderr)
            new_parts = comm.broadcast_object(decision)
        solver.barrier()
        with _StdoutToStderr():
            solver.close()
        if new_parts is not None:
            parts = [tuple(pr) for pr in new_parts]
        # (a new strip handle and with it -- native carrier -- a new RCCL communicator: ncclCommInitRank on one node takes
        # about a second, at most twice per run, well inside the supervisor's limit)
        solver = make_solver(parts)       # from the initial condition again
        if new_parts is None:
            break
    return solver


# --------------------------------------------------------------------------------------------------
def main():
    a = parse()
    if a.fast_leg:
        return fast_leg(a)
    if "WORLD_SIZE" not in os.environ and a.gpus > 1 and not a.child:
        raise SystemExit(launch_workers(a))       # no launcher around us: be the launcher
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.same_device:
        local = 0               # (the rehearsal: every rank on the one GPU there is)
    if a.gpus != world:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d (start `python bench.py --gpus N` without a launcher, "
                         "or with python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 "
                         "--master-port P bench.py --gpus N)" % (a.gpus, world))
    if a.dry_run:
        if os.environ.get("VOF2D_BENCH_TEST_DIE_RANK") == str(rank):      # self-test of the launcher's clean-up
            sys.exit(7)
        from vof2d.comms import EnvComm
        comm = EnvComm(rank, world, local)
        token = comm.broadcast_object({"token": os.getpid()} if rank == 0 else None)
        got = comm.gather_object({"rank": rank, "local_rank": local, "world": world, "token": token["token"], "same_device": bool(a.same_device),
                                  "master": "%s:%s" % (os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT"))})
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks": got}), flush=True)
            time.sleep(0.2)
            comm.cleanup()
        return
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
    api = None
    if not dist_path:
        # single GPU: no torch in the process at all -- ctypes -> C ABI -> HIP
        from vof2d._lib import hip_api
        from vof2d.engine import Engine, make_desc
        api = hip_api()
        eng = Engine(api, make_desc(api, nx, ny, a.dtype, "f32", device=local, jacobi_iters=a.jacobi_iters, dt=dt))
        eng.set_init_F(a.ic)
        handle_warm_ms, warm_violations = warm_handle(eng, a.ic, nx, ny, a.dtype)
        elapsed = timed_steps(eng, a.warmup, a.steps)
        # (read now: the knob changes of the Jacobi timing below make the handle forget its choice)
        form_state = (eng.get_param("overlap_halves"), eng.get_counter("tm_choice"), eng.get_counter("tm_steps"), eng.get_param("gas_share"))
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
            solver = balance_strips(a, comm, solver, make_solver, rank, world, nx)
            native_ok = True
        except Exception as exc:   # e.g. no loadable RCCL: symmetric on all ranks -> the torch carrier
            print("[bench] native RCCL exchange unavailable (%r); falling back to torch.distributed" % (exc,), file=sys.stderr)
    graph_steps = None
    if not dist_path:
        pass
    elif native_ok:
        with _StdoutToStderr():
            eng = solver.eng
            solver.step(a.warmup, overlap=a.overlap)
            eng.sync()
            solver.barrier()
        g0 = eng.get_counter("exchange_graph_steps") if world > 1 else 0
        t0 = time.perf_counter()
        solver.step(a.steps, overlap=a.overlap)
        eng.sync()          # the compute stream has joined the communication stream of every step
        solver.barrier()
        elapsed = time.perf_counter() - t0
        elapsed = comm.allreduce_max(elapsed, eng)
        graph_steps = (eng.get_counter("exchange_graph_steps") - g0) if world > 1 else None
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
        # the second carrier: torch.distributed P2P.  Backend "nccl" (= RCCL) between GPUs; "gloo" with the rows staged through
        # host memory for the one-GPU rehearsal (--same-device), where two ranks share a device and RCCL cannot be used at all
        gloo = bool(a.same_device)
        torch.cuda.set_device(local)
        ov = a.overlap if a.overlap == 5 else bool(a.overlap)     # (5: the library's pair kernels piece by piece, StripSolver._step_pieces)
        with _StdoutToStderr():
            if gloo:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))

            def make_solver_t(parts=None):
                return StripSolver(nx, ny, a.dtype, ic=a.ic, rank=rank, world=world, device=local, parts=parts,
                                   jacobi_iters=a.jacobi_iters, exchange="torch", dt=dt)
            solver = make_solver_t()
            if world > 1:
                solver = balance_strips(a, solver.comm, solver, make_solver_t, rank, world, nx)
            eng = solver.eng
            solver.step(a.warmup, overlap=ov)
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver.step(a.steps, overlap=ov)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if gloo else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        exchange = ("torch/gloo (host-staged)" if gloo else "torch") if world > 1 else "none"

    digests = None
    if dist_path and a.digest:
        import hashlib
        eng.sync()          # (before the Jacobi timing below advances p)
        got = {f: solver.gather(f) for f in ("F", "u", "v", "p")}       # (every rank takes part; rank 0 holds the fields)
        if rank == 0:
            digests = {f: hashlib.sha256(got[f].tobytes()).hexdigest() for f in got}
            digests["istep"] = int(eng.istep)
            digests["shape"] = list(got["F"].shape)
        del got
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
    eng.set_param("solve_pairs", 0)            # (this record is k_jacobi_tb's: five sweeps per launch, not the pair kernel's ten)
    ms_sweep_tb = eng.time_jacobi(nt)
    eng.set_param("jacobi_tb", 1)
    ms_sweep_1 = eng.time_jacobi(max(2, a.jacobi_sweeps_timed // 2 * 2))
    eng.set_param("jacobi_tb", tb)
    violations = eng.get_counter("courant_violations") - (warm_violations if not dist_path else 0)
    achieved_1 = sweep_bytes / (ms_sweep_1 * 1e-3) / 1e9
    traffic, traffic_note = load_pmc_traffic(nx, ny, a.dtype) if not dist_path else ({}, "N > 1")
    fused = {"kernel": "k_jacobi_tb", "sweeps_per_launch": tb, "bound": "hbm (actual traffic: lead-in rows + tile overlap)",
             "us_per_launch_back_to_back": 1e3 * ms_sweep_tb * tb, "us_per_sweep_back_to_back": 1e3 * ms_sweep_tb,
             "hbm_traffic_bytes_per_launch": traffic.get("tb")}
    if not dist_path and a.jacobi_iters > 0 and a.jacobi_iters % tb == 0:
        # in-step cost: (step with sweeps - step without sweeps) / launches, both graph-replayed
        from vof2d.engine import Engine as _E, make_desc as _md
        e0 = _E(api, _md(api, nx, ny, a.dtype, "f32", device=local, jacobi_iters=0, dt=dt))
        e0.set_init_F(a.ic)
        ms0 = 1e3 * timed_steps(e0, a.warmup, a.steps) / a.steps
        e0.close()
        launches = a.jacobi_iters // tb
        us_in_step = (1e3 * elapsed / a.steps - ms0) * 1e3 / launches
        fused.update({"us_per_launch_in_step": us_in_step, "us_per_sweep_in_step": us_in_step / tb,
                      "ms_per_step_without_sweeps": ms0,
                      "algorithmic_GBs_in_step": sweep_bytes * tb / (us_in_step * 1e-6) / 1e9,
                      "frac_of_peak_on_its_3_passes": sweep_bytes / (us_in_step * 1e-6) / 1e9 / HBM_PEAK_GBS})
    # The N > 1 runs strong-scale 8192^2; give the single-GPU figure for that grid too, so the
    # scaling series has its own N = 1 point: in the N = 1 line (default workload only), and --
    # measured in the same job, on rank 0's GPU, after the timed region -- in every N > 1 line.
    ref8192 = None
    speedup = None
    if rank == 0 and not a.no_scaling_reference and ((not dist_path and not a.nx) or (world > 1 and nx == ny)):
        try:
            if api is None:
                from vof2d._lib import hip_api
                api = hip_api()
            nref = 8192 if not dist_path else nx
            ref8192 = single_gpu_reference(api, nref, a.dtype, a.ic, local, a.jacobi_iters, stable_dt(nref))
            if dist_path:
                speedup = (nx * ny * a.steps / elapsed) / ref8192["value"]
        except Exception as exc:   # e.g. not enough free HBM
            ref8192 = {"error": str(exc)}
    # the kernels of the step under the built-in profiler (one HIP event pair per dispatch, on the stream the kernels are
    # launched on), on a fresh run of the same workload, steps 11 .. 10 + --profile-steps: long enough to contain the
    # regime in which the tiny-value front of the pressure iteration crosses the grid.  (`eng` has just had its p
    # advanced by the back-to-back Jacobi launches above, which is not a state the solver passes through.)
    prof, prof_note, nprof = {}, None, max(1, a.profile_steps)
    prof_run = {}
    if not dist_path:
        try:
            from vof2d.engine import Engine as _E, make_desc as _md
            e1 = _E(api, _md(api, nx, ny, a.dtype, "f32", device=local, jacobi_iters=a.jacobi_iters, dt=dt))
            try:
                e1.set_init_F(a.ic)
                e1.step(10)
                prof = e1.profile_steps(nprof)
                # ... and the launches of the form the handle keeps (k_tm / k_jacobi_pair on a large fp64 grid), the same
                # steps of the same run: the handle is warmed (which decides the form) and put back to the initial state
                kept_tm = form_state[1] == 1 or (form_state[1] < 0 and form_state[2] > 0)
                if kept_tm:
                    e1.set_param("fuse_tm", 1)     # (the form the timed handle runs)
                    warm_handle(e1, a.ic, nx, ny, a.dtype)
                    e1.step(10)
                    prof_run = e1.profile_steps(nprof)
            finally:
                e1.close()
        except Exception as exc:      # e.g. not enough free HBM for a second engine: the line survives without the breakdown
            prof, prof_note = {}, "in-situ profile failed: %s" % (exc,)
    kernels_us = {k: round(v[0], 2) for k, v in prof.items()}
    cells = nx * ny
    step_kernels = {k: {"launches_per_step": round(v[1] / float(nprof), 2), "algorithmic_passes": KERNEL_PASSES[k],
                        "algorithmic_bytes_per_launch": KERNEL_PASSES[k] * esz * cells, "us_per_launch_dispatch": round(v[0], 2),
                        "frac_of_peak": KERNEL_PASSES[k] * esz * cells / (v[0] * 1e-6) / 1e9 / HBM_PEAK_GBS}
                    for k, v in prof.items() if k in KERNEL_PASSES and v[0] > 0}

    run_kernels = {k: {"launches_per_step": round(v[1] / float(nprof), 3), "algorithmic_passes": KERNEL_PASSES[k],
                       "algorithmic_bytes_per_launch": KERNEL_PASSES[k] * esz * cells, "us_per_launch_dispatch": round(v[0], 2),
                       "frac_of_peak": KERNEL_PASSES[k] * esz * cells / (v[0] * 1e-6) / 1e9 / HBM_PEAK_GBS}
                   for k, v in prof_run.items() if k in KERNEL_PASSES and v[0] > 0}

    # which schedule actually ran: the 19-pass one-kernel transport needs a full domain, or mode 4 with
    # EVERY timed step replayed from the captured exchange graph (an RCCL that cannot capture, or
    # VOF2D_XCHG_GRAPH=0, silently runs mode 4 as eager mode 1 with the two-kernel transport)
    effective_overlap = a.overlap if dist_path else None
    try:
        if exchange == "native":
            # (mode 5: the middle steps of a call are replayed, two per launch; its head, its tail and an odd middle step are eager)
            captured = eng.comm_info()[1] == 1 and (graph_steps == a.steps if a.overlap != 5 else graph_steps >= a.steps - 3)
            if a.overlap == 4 and not captured:
                effective_overlap = 1
            one_kernel_transport = a.overlap in (4, 5) and (captured or a.overlap == 5)
        else:
            one_kernel_transport = bool(eng.get_param("fuse_transport")) and exchange == "none"
    except Exception:
        one_kernel_transport = False
    ARRAYS_PER_STEP = ARRAYS_PER_STEP_FULL if one_kernel_transport else ARRAYS_PER_STEP_STRIP
    if dist_path and world > 1 and a.overlap == 5:
        ARRAYS_PER_STEP = 11        # k_jacobi_pair 3 + k_tm 8 per middle step (the head and the tail of a call: once per call)
    # What the step AS THE HANDLE RAN IT has to move, by SURVEY 8d's rule applied to its own kernel list (every distinct
    # array a kernel reads or writes, once, times its launches per step, from the in-situ profile of the kept form):
    # 8 x 15/16 (k_tm) + 10 x 1/16 (the last k_tm of a 16-step batch, which also stores u and v: the batches chain, no plain
    # k_momentum / k_transport is left) + 3 (k_jacobi_pair: ten sweeps) = 11.1 passes where the pair kernels run,
    # 6 + 2 x 3 + 7 = 19 for the four-kernel schedule.  Three
    # step fractions stand side by side in `config`: on this list, on the 19 passes of the four-kernel schedule every
    # earlier round quoted, and on the HBM traffic the counters saw.
    own_kernels = run_kernels if "k_tm" in run_kernels else step_kernels
    own_passes = sum(v["launches_per_step"] * v["algorithmic_passes"] for v in own_kernels.values()) if own_kernels else float(ARRAYS_PER_STEP)
    step_s = elapsed / a.steps
    traffic_all, traffic_all_note = ({}, "N > 1")
    if not dist_path:
        traffic_all, traffic_all_note = load_pmc_traffic(nx, ny, a.dtype, os.path.join(ROOT, "profiles", "tm_pmc.json" if "k_tm" in run_kernels else "jacobi_pmc.json"))
    fam = {"k_tm": "tm", "k_tm_uv": "tm_uv", "k_jacobi_pair": "pair", "k_jacobi_tb": "tb", "k_momentum": "momentum", "k_transport": "transport", "k_jacobi": "single"}
    step_traffic = None
    if own_kernels and all(fam.get(k) in traffic_all for k in own_kernels):
        step_traffic = sum(v["launches_per_step"] * traffic_all[fam[k]] for k, v in own_kernels.items())
    fractions = {
        "own_kernel_list": own_passes * esz * cells / step_s / 1e9 / HBM_PEAK_GBS,
        "reference_schedule_19_passes": ARRAYS_PER_STEP * esz * cells / step_s / 1e9 / HBM_PEAK_GBS,
        "counter_traffic": (step_traffic / step_s / 1e9 / HBM_PEAK_GBS) if step_traffic else None,
        "counter_traffic_note": traffic_all_note if step_traffic else ("no committed PMC passes for every kernel of the step (%s)" % traffic_all_note),
    }
    if rank == 0 and dist_path:
        # which carrier moved the halos, BEFORE the line (a reader of the log sees it even if the line is lost)
        print("[bench] carrier: %s exchange, overlap mode %s%s, %s" % (
            exchange, a.overlap, "" if effective_overlap == a.overlap else " requested but ran as mode %s" % effective_overlap,
            ("%s of %d timed steps replayed from the captured exchange graph" % (graph_steps, a.steps)) if graph_steps is not None
            else "eager launches"), file=sys.stderr, flush=True)
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
                "%d row strips, %d-row deep halo, per-field RCCL send/recv (%s) overlap mode %d%s" % (
                    world, solver.halo, exchange, a.overlap,
                    "" if effective_overlap == a.overlap else " requested, ran as mode %d (exchange not captured)" % effective_overlap)),
                "nx": nx, "ny": ny, "dt": dt, "jacobi_iters": a.jacobi_iters,
                "exchange": exchange, "overlap": a.overlap if dist_path else None,
                "overlap_effective": effective_overlap,
                "rows_per_rank": [hi - lo + 1 for lo, hi in solver.parts] if dist_path else None,
                "exchange_graph": (eng.comm_info()[1] == 1) if exchange == "native" else None,
                "exchange_graph_steps_in_timed_region": graph_steps,
                "multi_gpu_hardware_verified": False if world > 1 else None,
                # the one-GPU rehearsal of the N > 1 path (--same-device): every rank on device 0, halos over gloo through host
                # memory -- the line's rate is of N processes sharing one GPU and says nothing about scaling
                "same_device_rehearsal": bool(a.same_device) if dist_path else None,
                "arrays_per_cell_update": round(own_passes, 3),
                "bytes_per_cell_update_algorithmic": round(own_passes * esz, 2),
                "arrays_per_cell_update_four_kernel_schedule": ARRAYS_PER_STEP,
                "step_kernel_list": {k: {"launches_per_step": v["launches_per_step"], "passes": v["algorithmic_passes"]} for k, v in own_kernels.items()} or None,
                "step_frac_of_peak": fractions,
                "tm_choice": form_state[1] if not dist_path else None,      # 1 / 0: the rule of vof_step chose k_tm / the chains; -1: the rule does not apply here
                "gas_share": form_state[3] if not dist_path else None,      # what the rule saw: the share of exact-zero cells of F
                # two-chain: every kernel of a step as two launches (rows above / below a moving boundary) on two
                # streams, the lower chain one kernel behind the upper; one-chain: four launches per
                # step, one after the other (small grids, strips, VOF2D_OVERLAP_HALVES=0)
                "step_schedule": _batch_form(form_state[:3]) if not dist_path else "strips",
                "handle_warm_steps": HANDLE_WARM_STEPS if not dist_path else None,
                "handle_warm_ms": round(handle_warm_ms, 2) if not dist_path else None},
            # `roofline` = the Poisson Jacobi kernel THE STEP RUNS, k_jacobi_tb (five sweeps per launch): algorithmic bytes =
            # 3 arrays x sizeof(T) x cells per launch (read p, read rhs, write p after five sweeps; SURVEY 8d's 24 B rule
            # per launch), duration = its average dispatch over the in-situ profile above (HIP events on the launch
            # stream, steps 11 .. 10 + --profile-steps of a fresh run, the tiny-value front included); `traffic` = HBM
            # bytes per launch from the committed rocprofv3 --pmc passes (profiles/jacobi_pmc.json).  The single-sweep
            # kernel of the north star's wording (k_jacobi, not launched by the step) is `north_star_single_sweep`.
            "roofline": roofline_record(step_kernels, traffic, traffic_note, nprof, prof_note, sweep_bytes, fused, tb, {
                "kernel": "k_jacobi", "bound": "hbm", "achieved": achieved_1, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved_1 / HBM_PEAK_GBS, "traffic": traffic.get("single"), "us_per_launch": 1e3 * ms_sweep_1,
                "algorithmic_bytes_per_launch": sweep_bytes, "launches_timed": max(2, a.jacobi_sweeps_timed // 2 * 2),
                "note": "one sweep per launch, back-to-back launches between one HIP event pair (vof_time_jacobi); used by "
                        "the step only for sweep counts that are not a multiple of five and for the residual checks"}),
            "jacobi_fused": fused,
            "step_kernels_as_run": run_kernels or None,
            # the kernels the step itself runs, same counting rule (built-in profiler: dispatch start ->
            # stop; reads a few us high behind a long-tailed predecessor -- rocprofv3, profiles/, is the reference)
            "step_kernels": step_kernels,
            "strong_scaling_reference_n1": ref8192,
            "fields_sha256": digests,
            "speedup_same_grid": speedup,
            "step_hbm_gbs_algorithmic": own_passes * esz * nx * ny * a.steps / elapsed / 1e9,
            "step_frac_of_peak_algorithmic": fractions["own_kernel_list"],
            "step_frac_of_peak_on_the_four_kernel_schedule": fractions["reference_schedule_19_passes"],
            "step_frac_of_peak_counter_traffic": fractions["counter_traffic"],
            "kernels_us_dispatch_start_to_stop": kernels_us,
            "courant_violations": violations,
        }
        if "k_tm" in run_kernels:
            traffic_tm, note_tm = load_pmc_traffic(nx, ny, a.dtype, os.path.join(ROOT, "profiles", "tm_pmc.json"))
            out["roofline"] = roofline_of_the_kept_form(run_kernels, out["roofline"], traffic_tm, note_tm, nprof, esz, cells)
        if not dist_path and not a.no_extras:
            try:
                out["sustained"] = sustained_record(api, nx, ny, a.dtype, a.ic, local, a.jacobi_iters, dt, a.sustained_steps)
                out["config"]["sustained_ms_per_step"] = round(out["sustained"]["ms_per_step"], 4)     # (in `config` too: the driver's record keeps that object)
                out["config"]["sustained_steps"] = out["sustained"]["steps"]
            except Exception as exc:
                out["sustained"] = {"error": str(exc)}
            if a.ic == 1:
                try:
                    wc = worst_case_record(api, nx, ny, a.dtype, local, a.jacobi_iters, dt, a.steps)
                    wc["slowdown_vs_the_headline_workload"] = wc["ms_per_step"] / (1e3 * elapsed / a.steps)
                    out["config"]["worst_case"] = wc
                except Exception as exc:
                    out["config"]["worst_case"] = {"error": str(exc)}
            try:   # what bit-faithful arithmetic costs: the same workload on the FMA-contracted build (own process)
                out["fast_build"] = run_fast_leg(a)
                if "value" in out["fast_build"]:
                    out["fast_build"]["speedup_over_parity_build"] = out["fast_build"]["value"] / out["value"]
            except Exception as exc:
                out["fast_build"] = {"error": str(exc)}
            if not a.nx:      # default workload only
                try:
                    out["residual_solve_1024"] = residual_solve_1024(api, local)
                except Exception as exc:
                    out["residual_solve_1024"] = {"error": str(exc)}
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
