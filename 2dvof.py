#!/usr/bin/env python3
"""2dvof.py -- drop-in command line of the reference solver, on the MI355X HIP library.

Same flags as /root/reference/2dvof.py:11-17 (`-ic {1,2,3}`, `-s`), same banner (:95-99), same
per-100-step status line (:533), same `output/NNNNNN-f.png` naming and plot (:563-571), same
`output/` and `data/` directories (:500-501).  The Taichi GUI is replaced by a headless loop;
everything numerical happens in libvof2d_hip.so (one C-ABI call per reference kernel).

Extensions (not in the reference, which hard-codes them at :9,:19-20 and loops until 'q'):
    --nx/--ny        grid size            (default 200 x 200)
    --dtype          f32 | f64            (default f32 = ti.f32)
    --steps N        stop after N steps   (default 0 = run until Ctrl-C, like the reference)
    --verbs          call the kernels one by one exactly as the main loop :513-528 does, instead of
                     the fused vof_step() schedule (same results, more HBM traffic)
    --jacobi-tol T   residual-terminated pressure solve, capped by --jacobi-max sweeps;
    --jacobi-crit    abs: max|p_new-p| <= T (default), rel: max|p_new-p| / max|p_new| <= T (vof_solve_p)
    --coord-cast     f32 | none           (keep / drop the .astype(np.float32) of :43,:45)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "taichi-2d-vof_amd"))

parser = argparse.ArgumentParser()  # Get the initial condition
# 1 - Dam Break; 2 - Rising Bubble; 3 - Droping liquid
parser.add_argument('-ic', type=int, choices=[1, 2, 3], default=1)
parser.add_argument('-s', action='store_true')
parser.add_argument('--nx', type=int, default=200)
parser.add_argument('--ny', type=int, default=200)
parser.add_argument('--dtype', choices=['f32', 'f64'], default='f32')
parser.add_argument('--coord-cast', choices=['f32', 'none'], default='f32')
parser.add_argument('--steps', type=int, default=0)
parser.add_argument('--verbs', action='store_true')
parser.add_argument('--jacobi-tol', type=float, default=0.0)
parser.add_argument('--jacobi-max', type=int, default=2000)
parser.add_argument('--jacobi-crit', choices=['abs', 'rel'], default='abs')
parser.add_argument('--device', type=int, default=0)
parser.add_argument('--vis', type=int, choices=[0, 1, 2, 3, 4], default=0,
                    help='what the reference GUI would display (SPACE cycles it there, 2dvof.py:508-509): '
                         '0 VOF, 1 u, 2 v, 3 |velocity|, 4 velocity vectors; saved as output/NNNNNN-vis.png with -s')


def main():
    args = parser.parse_args()
    from vof2d import VOF2D

    initial_condition = args.ic
    SAVE_FIG = args.s
    sim = VOF2D(args.nx, args.ny, dtype=args.dtype, coord_cast=args.coord_cast, device=args.device)
    nx, ny, dt = args.nx, args.ny, sim.dt
    rho_l, rho_g = sim.eng.get_param("rho_l"), sim.eng.get_param("rho_g")
    nu_l, nu_g = sim.eng.get_param("nu_l"), sim.eng.get_param("nu_g")
    gy = sim.eng.get_param("gy")
    Lx, Ly = sim.eng.get_param("Lx"), sim.eng.get_param("Ly")

    print(f'>>> A VOF solver written in HIP for MI355X; Press Ctrl-C to exit.')
    print(f'>>> Grid resolution: {nx} x {ny}, dt = {dt:4.2e}')
    print(f'>>> Density ratio: {rho_l / rho_g : 4.2f}, gravity : {gy : 4.2f}, sigma : {sim.sigma[None] : 4.2f}')
    print(f'>>> Viscosity ratio: {nu_l / nu_g : 4.2f}')

    istep = 0
    nstep = 100  # Interval to update output
    sim.set_init_F(initial_condition)
    os.makedirs('output', exist_ok=True)  # Make dir for output
    os.makedirs('data', exist_ok=True)  # Make dir for data save; only used for debugging

    def advance(n):
        if args.jacobi_tol > 0.0:
            for _ in range(n):   # main loop :513-528 with the residual-terminated solve (extension)
                sim.istep = sim.istep + 1
                sim.cal_nu_rho(); sim.get_normal_young(); sim.advect_upwind(); sim.set_BC()
                sim.eng.solve_p(args.jacobi_tol, args.jacobi_max, 10, args.jacobi_crit)
                sim.update_uv(); sim.set_BC()
                sim.solve_VOF_rudman(sim.istep); sim.post_process_f(); sim.set_BC()
        elif args.verbs:
            sim.step_verbs(n)
        else:
            sim.step(n)

    try:
        while args.steps == 0 or istep < args.steps:
            n = nstep - istep % nstep
            if args.steps:
                n = min(n, args.steps - istep)
            advance(n)
            istep += n
            if (istep % nstep) == 0:  # Output data every <nstep> steps
                warn = sim.courant_violations
                what = ('VOF field', 'u velocity', 'v velocity', 'velocity norm', 'velocity vectors')[args.vis]
                print(f'>>> Number of steps:{istep:<5d}, Time:{istep*dt:5.2e} sec. Displaying {what}.'
                      + (f' [{warn} Courant warnings]' if warn else ''))
                if SAVE_FIG:
                    import matplotlib
                    matplotlib.use('Agg')
                    import matplotlib.pyplot as plt
                    import matplotlib.cm as cm
                    count = istep // nstep - 1
                    # what gui.set_image(...) shows in the reference (:531-559), written to a file
                    if args.vis == 4:
                        V = sim.interp_velocity()
                        sp = max(4, nx // 50)
                        plt.figure(figsize=(5, Ly / Lx * 5))
                        plt.axis('off')
                        plt.quiver(V[1:nx + 1:sp, 1:ny + 1:sp, 0].T, V[1:nx + 1:sp, 1:ny + 1:sp, 1].T)
                        plt.savefig(f'output/{count:06d}-vis.png')
                        plt.close()
                    else:
                        img = (sim.get_vof_field, sim.get_u_field, sim.get_v_field, sim.get_vnorm_field)[args.vis]()
                        cmap = (cm.Blues, cm.coolwarm, cm.coolwarm, cm.plasma)[args.vis]
                        plt.imsave(f'output/{count:06d}-vis.png', cmap(img.transpose(1, 0)[::-1]))
                    Fnp = sim.F.to_numpy()
                    fx, fy = 5, Ly / Lx * 5
                    plt.figure(figsize=(fx, fy))
                    plt.axis('off')
                    plt.contourf(Fnp.T, cmap=plt.cm.Blues)
                    plt.savefig(f'output/{count:06d}-f.png')
                    plt.close()
    except KeyboardInterrupt:
        pass
    sim.sync()
    sim.close()


if __name__ == '__main__':
    main()
