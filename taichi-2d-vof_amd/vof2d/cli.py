"""The command line of 2dvof.py on the HIP library: argument parsing and the main loop.

Same flags as /root/reference/2dvof.py:11-17 (`-ic {1,2,3}`, `-s`), same banner (:95-99), same
per-100-step status line (:533-557), same `output/NNNNNN-f.png` naming and plot (:563-571), same
`output/` and `data/` directories (:500-501).  The Taichi GUI is replaced by a headless loop.

`run(args, api=None, comm=None)` is the whole program for one rank; `2dvof.py` calls it with the
HIP library (the only engine the product has).  Tests hand it another object with the same C ABI
and a torch.distributed (gloo) carrier to exercise the multi-rank host logic without a GPU.
"""
import argparse
import os
import sys

import numpy as np

STATE = ("F", "u", "v", "p")


def build_parser():
    parser = argparse.ArgumentParser(
        description="2-D VOF solver (drop-in command line of taichi-2d-vof's 2dvof.py on MI355X)")
    # 1 - Dam Break; 2 - Rising Bubble; 3 - Droping liquid          (2dvof.py:12-14)
    parser.add_argument('-ic', type=int, choices=[1, 2, 3], default=1)
    parser.add_argument('-s', action='store_true')
    ext = parser.add_argument_group("extensions (the reference hard-codes these at :9, :19-20, :521 and loops until 'q')")
    ext.add_argument('--nx', type=int, default=200)
    ext.add_argument('--ny', type=int, default=200)
    ext.add_argument('--dtype', choices=['f32', 'f64'], default='f32', help='field precision (default f32 = ti.f32, :9)')
    ext.add_argument('--coord-cast', choices=['f32', 'none'], default='f32',
                     help='keep / drop the .astype(np.float32) of the mesh coordinates (:43, :45)')
    ext.add_argument('--steps', type=int, default=0, help='stop after N steps (default 0: run until Ctrl-C, like the reference)')
    ext.add_argument('--dt', type=float, default=None, help='time step (default 4e-6, :33)')
    ext.add_argument('--jacobi-iters', type=int, default=10, help='pressure sweeps per step (reference: 10, :521)')
    ext.add_argument('--gpus', type=int, default=1,
                     help='row strips over N GPUs of this node, one process per GPU, halo exchange over RCCL '
                          '(started by this command itself, or by torch.distributed.run)')
    ext.add_argument('--verbs', action='store_true',
                     help='call the kernels one by one exactly as the main loop :513-528 does, instead of the fused '
                          'vof_step() schedule (same results, more HBM traffic; one GPU)')
    ext.add_argument('--jacobi-tol', type=float, default=0.0,
                     help='residual-terminated pressure solve instead of the fixed sweep count, capped by --jacobi-max '
                          '(with --gpus N the two norms are all-reduced over the strips)')
    ext.add_argument('--jacobi-max', type=int, default=2000)
    ext.add_argument('--jacobi-crit', choices=['abs', 'rel'], default='abs',
                     help='abs: max|p_new-p| <= tol (default); rel: max|p_new-p| / max|p_new| <= tol (vof_solve_p)')
    ext.add_argument('--device', type=int, default=0, help='HIP device ordinal (one GPU; with --gpus N rank r takes device r)')
    ext.add_argument('--vis', type=int, choices=[0, 1, 2, 3, 4], default=0,
                     help='what the reference GUI would display (SPACE cycles it there, :508-509): 0 VOF, 1 u, 2 v, '
                          '3 |velocity|, 4 velocity vectors; saved as output/NNNNNN-vis.png with -s (one GPU)')
    ext.add_argument('--save-every', type=int, default=0, metavar='N',
                     help='write data/NNNNNNNN.npz (F, u, v, p with ghost cells, istep) every N steps: the use the '
                          "reference's data/ directory (:501) was made for")
    ext.add_argument('--resume', default=None, metavar='FILE', help='continue from a file written by --save-every')
    return parser


def numerics_of(args, dt):
    """What, besides the fields, decides how a run continues: a checkpoint resumed with other values of these is
    a different run, not a continuation."""
    return {"dt": float(dt), "jacobi_iters": int(args.jacobi_iters), "coord_cast": str(args.coord_cast),
            "jacobi_tol": float(args.jacobi_tol), "jacobi_max": int(args.jacobi_max) if args.jacobi_tol > 0.0 else 0,
            "jacobi_crit": str(args.jacobi_crit) if args.jacobi_tol > 0.0 else ""}


def save_state(path, fields, istep, nx, ny, dtype, ic, courant=0, numerics=None):
    tmp = path + ".tmp.npz"
    extra = {"num_" + k: np.array(v) for k, v in (numerics or {}).items()}
    np.savez(tmp, istep=np.int64(istep), nx=np.int64(nx), ny=np.int64(ny), dtype=np.array(dtype), ic=np.int64(ic),
             courant_violations=np.int64(courant), **extra, **fields)
    os.replace(tmp, path)


def load_state(path, nx, ny, dtype, numerics=None):
    """(fields, istep, courant_violations).  Refuses a file of another grid / precision, and one written with other
    numerics (dt, sweep count, coordinate cast, residual criterion) than this run's: the tests promise an exact
    continuation.  Files from before the numerics were recorded are accepted as they are."""
    z = np.load(path, allow_pickle=False)
    if (int(z["nx"]), int(z["ny"]), str(z["dtype"])) != (nx, ny, dtype):
        raise SystemExit("--resume: %s holds a %dx%d %s run, this one is %dx%d %s" %
                         (path, int(z["nx"]), int(z["ny"]), str(z["dtype"]), nx, ny, dtype))
    for k, v in (numerics or {}).items():
        key = "num_" + k
        if key in z.files and z[key].item() != v:
            raise SystemExit("--resume: %s was written with %s = %r, this run has %r (pass the same value to continue it)" %
                             (path, k.replace("_", "-"), z[key].item(), v))
    return {f: z[f] for f in STATE}, int(z["istep"]), int(z["courant_violations"]) if "courant_violations" in z.files else 0


class _Single:
    """One GPU: the VOF2D mirror of the reference's fields and kernels."""

    def __init__(self, args, api):
        from .solver import VOF2D
        consts = {} if args.dt is None else {"dt": args.dt}
        self.sim = VOF2D(args.nx, args.ny, dtype=args.dtype, coord_cast=args.coord_cast, device=args.device,
                         jacobi_iters=args.jacobi_iters, api=api, **consts)
        self.eng, self.rank, self.args = self.sim.eng, 0, args
        self.base_courant = 0

    def init(self, ic):
        self.sim.set_init_F(ic)

    def load(self, fields, istep, courant=0):
        for f in STATE:
            self.eng.set(f, fields[f])
        self.eng.istep = istep
        self.base_courant = courant                  # (the device counter restarts at zero)

    def advance(self, n):
        a, sim = self.args, self.sim
        if a.jacobi_tol > 0.0:
            for _ in range(n):   # main loop :513-528 with the residual-terminated solve (extension)
                sim.istep = sim.istep + 1
                sim.cal_nu_rho(); sim.get_normal_young(); sim.advect_upwind(); sim.set_BC()
                self.eng.solve_p(a.jacobi_tol, a.jacobi_max, 10, a.jacobi_crit)
                sim.update_uv(); sim.set_BC()
                sim.solve_VOF_rudman(sim.istep); sim.post_process_f(); sim.set_BC()
        elif a.verbs:
            sim.step_verbs(n)
        else:
            sim.step(n)

    def full(self, name):
        return self.eng.get(name)

    def courant(self):
        return self.base_courant + self.sim.courant_violations

    def close(self):
        self.sim.sync()
        self.sim.close()


class _Strips:
    """N GPUs: this rank's row strip (vof2d/strips.py); rank 0 gathers what it writes to disk."""

    def __init__(self, args, api, comm, rank, world):
        from .strips import StripSolver
        consts = {} if args.dt is None else {"dt": args.dt}
        self.s = StripSolver(args.nx, args.ny, args.dtype, ic=args.ic, coord_cast=args.coord_cast,
                             jacobi_iters=args.jacobi_iters, rank=rank, world=world, device=rank, api=api,
                             comm=comm, dist=getattr(comm, "dist", None), **consts)
        self.eng, self.rank, self.world, self.args = self.s.eng, rank, world, args
        self.base_courant = 0

    def init(self, ic):
        pass                                         # StripSolver ran set_init_F on its strip

    def load(self, fields, istep, courant=0):
        lo, hi = self.s.rows
        for f in STATE:
            self.eng.set(f, fields[f][lo:hi + 1])   # every rank reads the file: owned rows and halos alike
        self.eng.istep = istep
        self.base_courant = courant                  # (the device counters restart at zero)

    def advance(self, n):
        a = self.args
        if a.jacobi_tol <= 0.0:
            self.s.step(n)
            return
        e = self.eng
        for _ in range(n):   # main loop :513-528, verb by verb on the extended strip, with the residual-terminated solve
            e.istep = e.istep + 1
            e.cal_nu_rho(); e.get_normal_young(); e.advect_upwind(); e.set_BC()
            # (StripSolver.solve_p: the sweeps of a check in batches the deep halo of p covers, p exchanged after every
            # batch, both norms all-reduced (MAX) -- same sweep counts and residuals as vof_solve_p on one domain)
            self.s.solve_p(a.jacobi_tol, a.jacobi_max, 10, a.jacobi_crit)
            e.update_uv(); e.set_BC()
            e.solve_VOF_rudman(e.istep); e.post_process_f(); e.set_BC()
            self.s.exchange()                         # F, u, v, p: the 8 rows the step consumed are well inside the halo

    def full(self, name):
        return self.s.gather(name)                   # rank 0: (nx+2, ny+2); others: None

    def courant(self):
        parts = self.s.comm.gather_object(int(self.eng.get_counter("courant_violations")))
        return self.base_courant + sum(parts) if self.rank == 0 else 0

    def close(self):
        self.s.barrier()
        self.s.close()


def run(args, api=None, comm=None, rank=None, world=None, out=None):
    """The program of one rank.  api: the C-ABI object (default: the HIP library; there is no other
    engine in the product).  comm / rank / world: the process group when several ranks run."""
    say = out if out is not None else (lambda *a: print(*a, flush=True))
    if world is None:
        env_world = int(os.environ.get("WORLD_SIZE", 1))
        if args.gpus == 1 and env_world > 1:
            # e.g. torchrun without --gpus: every rank would run the whole problem on device 0 and race on output/, data/
            raise SystemExit("started under a launcher with WORLD_SIZE = %d but --gpus is 1: pass --gpus %d (row strips), "
                             "or start a single process" % (env_world, env_world))
        world = env_world if args.gpus > 1 else 1
    if rank is None:
        rank = int(os.environ.get("RANK", 0)) if world > 1 else 0
    if world > 1 and (args.verbs or args.vis):
        raise SystemExit("--verbs and --vis run on one GPU (drop --gpus)")
    if world > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d ranks" % (args.gpus, world))
    if api is None:
        from ._lib import hip_api
        api = hip_api()
    if world > 1:
        if comm is None:
            from .comms import EnvComm
            comm = EnvComm()
        drv = _Strips(args, api, comm, rank, world)
    else:
        drv = _Single(args, api)
    eng = drv.eng
    nx, ny, dt = args.nx, args.ny, eng.get_param("dt")
    rho_l, rho_g = eng.get_param("rho_l"), eng.get_param("rho_g")
    nu_l, nu_g = eng.get_param("nu_l"), eng.get_param("nu_g")
    gy, sigma = eng.get_param("gy"), eng.get_param("sigma")
    Lx, Ly = eng.get_param("Lx"), eng.get_param("Ly")
    lead = rank == 0

    if lead:
        say(f'>>> A VOF solver written in HIP for MI355X; Press Ctrl-C to exit.')
        say(f'>>> Grid resolution: {nx} x {ny}, dt = {dt:4.2e}' + (f', {world} row strips' if world > 1 else ''))
        say(f'>>> Density ratio: {rho_l / rho_g : 4.2f}, gravity : {gy : 4.2f}, sigma : {sigma : 4.2f}')
        say(f'>>> Viscosity ratio: {nu_l / nu_g : 4.2f}')

    istep = 0
    nstep = 100  # Interval to update output                       (2dvof.py:497)
    drv.init(args.ic)
    if lead:
        os.makedirs('output', exist_ok=True)  # Make dir for output                 (:500)
        os.makedirs('data', exist_ok=True)    # Make dir for data save               (:501)
    numerics = numerics_of(args, dt)
    if args.resume:
        fields, istep, warn0 = load_state(args.resume, nx, ny, args.dtype, numerics)
        drv.load(fields, istep, warn0)
        if lead:
            say(f'>>> Resumed from {args.resume} at step {istep}.')

    def next_stop(i):
        n = nstep - i % nstep
        if args.save_every:
            n = min(n, args.save_every - i % args.save_every)
        if args.steps:
            n = min(n, args.steps - i)
        return n

    try:
        while args.steps == 0 or istep < args.steps:
            n = next_stop(istep)
            drv.advance(n)
            istep += n
            if args.save_every and istep % args.save_every == 0:
                fields = {f: drv.full(f) for f in STATE}
                warn = drv.courant()
                if lead:
                    save_state('data/%08d.npz' % istep, fields, istep, nx, ny, args.dtype, args.ic, warn, numerics)
            if (istep % nstep) == 0:  # Output data every <nstep> steps            (:530)
                warn = drv.courant()
                Fnp = drv.full("F") if args.s else None
                if not lead:
                    continue
                from .vis import OPTIONS, save_display
                say(f'>>> Number of steps:{istep:<5d}, Time:{istep*dt:5.2e} sec. Displaying {OPTIONS[args.vis][0]}.'
                    + (f' [{warn} Courant warnings]' if warn else ''))
                if args.s:
                    import matplotlib
                    matplotlib.use('Agg')
                    import matplotlib.pyplot as plt
                    count = istep // nstep - 1
                    if world == 1:   # what gui.set_image / gui.arrows show in the reference (:531-559), as a file
                        save_display(f'output/{count:06d}-vis.png', drv.sim, args.vis)
                    fx, fy = 5, Ly / Lx * 5
                    plt.figure(figsize=(fx, fy))
                    plt.axis('off')
                    plt.contourf(Fnp.T, cmap=plt.cm.Blues)
                    plt.savefig(f'output/{count:06d}-f.png')
                    plt.close()
    except KeyboardInterrupt:
        pass
    drv.close()
    return 0


def main(script, argv=None):
    """Entry of 2dvof.py: parse; with --gpus N and no launcher around us become the launcher
    (vof2d/launch.py: a GPU-free parent and N fresh workers, never a re-exec of a process that
    touched the GPU); otherwise run this rank."""
    argv = list(sys.argv[1:] if argv is None else argv)
    args = build_parser().parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1:
        from .launch import spawn_ranks, under_launcher
        if not under_launcher():
            return spawn_ranks(script, argv, args.gpus)
    return run(args)
