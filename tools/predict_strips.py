#!/usr/bin/env python3
"""What the first multi-GPU run should show, predicted on ONE GPU: every strip of an N-way row decomposition of the
bench's N > 1 workload (8192^2 fp64 dam-break, dt 1e-6) is built in turn and stepped with its halos refreshed from a
full-domain run after every step (device copies stand in for the RCCL send/recv), its own kernels timed on the device
(vof_timer_*: no waiting for neighbours in the figure) -- first on equal strips, then on the cost-balanced partition
bench.py --gpus N would cut from those costs (strips.balanced_partition).  The step of an N-GPU run ends with its
slowest rank, so

    predicted speedup(N) = single-GPU ms/step of the same grid / max over ranks (kernel ms/step of the rank's strip)

if the exchange hides under the transport of the inner rows as it does on the loopback (profiles/r03e_strip_exchange.md).

    python3 tools/predict_strips.py [--nx 8192] [--ranks 2,4,8] [--steps 12] > profiles/<tag>_strip_prediction.md
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "taichi-2d-vof_amd"))
STATE = ("F", "u", "v", "p")


MODE5 = ("F", "u_star", "v_star", "rhs", "p")


PARAMS = []     # --param knob=value: set on every STRIP handle (not on the full domain that feeds its halos)


def strip_costs_pairs(api, nx, dtype, dt, parts, W, steps, skip=3):
    """The same for overlap mode 5 (the strips run k_jacobi_pair and k_tm, vof_step_tm_piece): ms per MIDDLE step of each
    strip's own kernels, its halos of F, u*, v*, rhs, p refreshed after every step from a full domain driven through the
    same pieces."""
    from vof2d.engine import Engine, make_desc
    from vof2d.strips import stored_rows
    costs = []
    for r, own in enumerate(parts):
        full = Engine(api, make_desc(api, nx, nx, dtype, "f32", device=0, dt=dt))
        rows = stored_rows(nx, own, W)
        s = Engine(api, make_desc(api, nx, nx, dtype, "f32", rows=rows, own=own, device=0, dt=dt))
        for kv in PARAMS:
            s.set_param(kv.split("=")[0], float(kv.split("=")[1]))

        def refresh(fields):
            for f in fields:                     # what the neighbours would send
                if rows[0] < own[0]:
                    s.copy_rows_from(full, f, rows[0], own[0] - 1)
                if rows[1] > own[1]:
                    s.copy_rows_from(full, f, own[1] + 1, rows[1])
        for x in (full, s):
            x.set_init_F(1)
            x.step(1)
        refresh(STATE)
        for x in (full, s):
            x.step_tm_piece(0)
        refresh(("u_star", "v_star", "rhs"))
        ms = []
        for k in range(steps):
            s.timer_start()
            s.step_tm_piece(1)
            ms.append(s.timer_stop())
            full.step_tm_piece(1)
            refresh(MODE5)
        costs.append(sum(ms[skip:]) / len(ms[skip:]))
        s.close()
        full.close()
    return costs


def strip_costs(api, nx, dtype, dt, parts, W, steps, skip=3):
    """ms per step of each strip's own kernels (device timer), halos refreshed from a full-domain run."""
    from vof2d.engine import Engine, make_desc
    from vof2d.strips import stored_rows
    costs = []
    for r, own in enumerate(parts):
        full = Engine(api, make_desc(api, nx, nx, dtype, "f32", device=0, dt=dt))
        full.set_init_F(1)
        rows = stored_rows(nx, own, W)
        s = Engine(api, make_desc(api, nx, nx, dtype, "f32", rows=rows, own=own, device=0, dt=dt))
        s.set_init_F(1)
        ms = []
        for k in range(steps):
            s.timer_start()
            s.step(1)
            ms.append(s.timer_stop())
            full.step(1)
            for f in STATE:                      # what the neighbours would send
                if rows[0] < own[0]:
                    s.copy_rows_from(full, f, rows[0], own[0] - 1)
                if rows[1] > own[1]:
                    s.copy_rows_from(full, f, own[1] + 1, rows[1])
        costs.append(sum(ms[skip:]) / len(ms[skip:]))
        s.close()
        full.close()
    return costs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=8192)
    ap.add_argument("--ranks", default="2,4,8")
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--pairs", action="store_true", help="overlap mode 5: the strips run k_jacobi_pair and k_tm (middle steps of a call)")
    ap.add_argument("--param", action="append", default=[], help="knob=value set on every strip handle (e.g. tm_rows=52)")
    a = ap.parse_args()
    PARAMS.extend(a.param)
    costs_of = strip_costs_pairs if a.pairs else strip_costs
    from vof2d import _abi
    from vof2d._lib import hip_api
    from vof2d.engine import Engine, make_desc
    from vof2d.strips import partition, balanced_partition
    api = hip_api()
    nx, dt, W = a.nx, (4e-6 if a.nx <= 4096 else 1e-6), _abi.halo_rows(10)
    e = Engine(api, make_desc(api, nx, nx, a.dtype, "f32", device=0, dt=dt))
    e.set_init_F(1)
    e.step(5); e.sync()
    t0 = time.perf_counter(); e.step(30); e.sync()
    one = 1e3 * (time.perf_counter() - t0) / 30
    e.close()
    print("# Predicted strip costs of `bench.py --gpus N` (%dx%d %s dam-break, dt %g), measured strip by strip on one MI355X\n" % (nx, nx, a.dtype, dt))
    print("single GPU, same grid: **%.3f ms/step** (steps 6-35, graph-replayed)\n" % one)
    print("| N | partition | rows per rank | kernel ms/step per rank | slowest | predicted speedup = %.3f / slowest |" % one)
    print("|---|---|---|---|---|---|")
    for n in [int(x) for x in a.ranks.split(",")]:
        parts = partition(nx, n)
        c = costs_of(api, nx, a.dtype, dt, parts, W, a.steps)
        print("| %d | equal | %s | %s | %.3f | %.2f |" % (n, " ".join(str(hi - lo + 1) for lo, hi in parts), " ".join("%.3f" % x for x in c), max(c), one / max(c)))
        if max(c) > 1.015 * sum(c) / len(c):
            parts2 = balanced_partition(nx, parts, c, min_rows=W)
            c2 = costs_of(api, nx, a.dtype, dt, parts2, W, a.steps)
            print("| %d | balanced (as bench.py re-cuts it) | %s | %s | %.3f | %.2f |" % (n, " ".join(str(hi - lo + 1) for lo, hi in parts2), " ".join("%.3f" % x for x in c2), max(c2), one / max(c2)))
        sys.stdout.flush()
    if a.pairs:
        print("\nThe per-rank figures are the middle steps of an overlap-mode-5 call (steps 5-%d of the run): k_jacobi_pair on all"
              "\nstored rows and k_tm (this step's transport + the next step's momentum) on the owned rows as ONE launch, kernels only,"
              "\ndevice-timed.  vof_step_exchange runs k_tm as two concurrent launches (edge bands on the communication stream in front"
              "\nof the send / recv group, the other rows beside them); the head and the tail of a call (k_momentum; two k_jacobi_tb +"
              "\nk_transport) are paid once per call of n steps." % (a.steps + 1))
        return
    print("\nThe per-rank figures are the first steps of the run (steps 4-%d): kernels only, device-timed; the strips run the"
          "\ntwo-kernel transport here (vof_step on a strip handle), the RCCL path (vof_step_exchange mode 4) the fused one on the"
          "\nedge bands and the inner rows, which `profiles/r03e_strip_exchange.md` measured 5-6 %% faster for the interior strip of 8."
          "\nWhat the SCALE record should be compared with: speedup(N) within ~10 %% of the last column if xGMI transfers hide"
          "\nlike the loopback copies; markedly lower means exposed exchange (look at config.exchange_graph_steps_in_timed_region"
          "\nand config.overlap_effective in the bench line first)." % a.steps)


if __name__ == "__main__":
    main()
