"""Differential fuzz of the HIP library against the CPU oracle: random grids, constants, schedule knobs and call
sequences from a seed, every field compared value for value after every call.

The fixed-case parity tests (test_parity_gpu.py) pin each code path on grids chosen for it; this one draws the
combinations nobody chose -- odd sizes with forced batch forms, chunk lengths that do not divide the rows, knobs
changed between batches, fields overwritten between chained batches, verbs between fused steps -- and replays the
reference's semantics (the oracle) beside the library through the same ABI calls.

Three more legs: random strip decompositions (2-6 uneven strips, three drivers) against the single domain, library against library;
invalid calls straight through the ABI (refused with a status, the state untouched); and -- in tests/_loopback_worker.py fuzz --
the captured RCCL exchange on random looped-back strips.  What the campaigns found: profiles/r06_fuzz.md.

    VOF_FUZZ_SEED   first seed (default 20261003)       VOF_FUZZ_CASES   cases of the default test (default 36)
    python tests/test_fuzz_gpu.py --seed S --cases N [--seconds T] [--log FILE] [--large P] [--huge P] [--strips]
        a campaign outside pytest (same generators; --large / --huge: share of grids of 0.3-2 M / 4-9 M cells)
"""
import os
import sys
import time

import numpy as np
import pytest

if __name__ == "__main__":   # (campaign mode: the paths conftest.py sets up for pytest)
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p_ in (os.path.join(ROOT, "taichi-2d-vof_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT):
        if p_ not in sys.path:
            sys.path.insert(0, p_)

from util import STATE, diff_report, engine

SCRATCH = ("u_star", "v_star", "rhs")
STATS = {"completed": 0, "blew_up": 0, "refused": 0, "ops": 0}   # how the cases of this process ended
COVERAGE = ("tm_steps", "tm_chained_batches", "pair_launches", "halves_steps")   # counters of the library: cases that ran that form
SEED0 = int(os.environ.get("VOF_FUZZ_SEED", "20261003"))
NCASES = int(os.environ.get("VOF_FUZZ_CASES", "36"))

# knobs that change the schedule and never a value (include/vof2d.h) with the values drawn for them
KNOBS = {
    "fuse_tm": (-1, 0, 1, 1, 1),
    "jacobi_pair": (0, 1, 2),
    "overlap_halves": (-1, 0, 0, 1, 2, 3),
    "buffer_stores": (0, 1, 2, 3, 4, 5, 6, 7, 7),
    "virtual_ghosts": (0, 1, 1),
    "fuse_transport": (0, 1, 1, 1),
    "jacobi_tb": (1, 5, 5, 5),
    "jacobi_tb_adapt": (0, 1),
    "jacobi_tb_general": (0, 0, 1),
    "jacobi_tb_rows": (0, 0, 3, 7, 16, 29, 51),
    "momentum_rows": (0, 0, 1, 2, 5, 14, 33),
    "fctx_corr_rows": (0, 0, 1, 4, 9, 16),
    "tm_rows": (0, 0, 1, 3, 6, 16, 37, 64, 200),
    "jacobi_pair_rows": (0, 0, 1, 5, 12, 27, 43, 80, 300),
    "batch_steps": (4, 8, 16, 32),
    "pair_slow10": (10, 20, 32, 50),
    "tb_slow10": (10, 20, 36),
    "solve_pairs": (0, 1),
}


def _abuses():
    """Calls the ABI must refuse with a status code -- no crash, no change of state (every entry point returns an int, no
    exceptions or aborts across the ABI).  Each entry: name, f(api, handle, engine) -> status."""
    import ctypes as C
    buf = (C.c_double * 8)()
    out = C.c_double()
    i64 = C.c_int64()
    i32 = C.c_int32()
    return [
        ("get_rows below the stored rows", lambda api, h, e: api.get_rows(h, b"F", e.row_lo - 1, e.row_lo + 1, buf, 64)),
        ("get_rows above the stored rows", lambda api, h, e: api.get_rows(h, b"p", e.row_hi, e.row_hi + 3, buf, 64)),
        ("get_rows reversed", lambda api, h, e: api.get_rows(h, b"u", 2, 1, buf, 64)),
        ("get_rows with a short buffer", lambda api, h, e: api.get_rows(h, b"v", 0, 1, buf, 8)),
        ("get_rows into NULL", lambda api, h, e: api.get_rows(h, b"F", 0, 0, None, (e.ny + 2) * 8)),
        ("get_field of an unknown name", lambda api, h, e: api.get_field(h, b"vorticity", buf, 64)),
        ("get_field with a NULL name", lambda api, h, e: api.get_field(h, None, buf, 64)),
        ("set_rows with a wrong size", lambda api, h, e: api.set_rows(h, b"F", 0, 0, buf, 24)),
        ("set_field from NULL", lambda api, h, e: api.set_field(h, b"u", None, 0)),
        ("step(-1)", lambda api, h, e: api.step(h, -1)),
        ("solve_p_jacobi(-3)", lambda api, h, e: api.solve_p_jacobi(h, -3)),
        ("set_init_F(7)", lambda api, h, e: api.set_init_F(h, 7)),
        ("set_init_F(0)", lambda api, h, e: api.set_init_F(h, 0)),
        ("step_phase(1) out of order", lambda api, h, e: api.step_phase(h, 1)),
        ("step_phase(5)", lambda api, h, e: api.step_phase(h, 5)),
        ("set_param of an unknown name", lambda api, h, e: api.set_param(h, b"viscosity", 1.0)),
        ("set_param with a NULL name", lambda api, h, e: api.set_param(h, None, 1.0)),
        ("get_param of an unknown name", lambda api, h, e: api.get_param(h, b"gamma", C.byref(out))),
        ("get_counter of an unknown name", lambda api, h, e: api.get_counter(h, b"launches", C.byref(i64))),
        ("get_vis_field of an unknown image", lambda api, h, e: api.get_vis_field(h, b"pressure", buf, 64)),
        ("get_vis_field with a short buffer", lambda api, h, e: api.get_vis_field(h, b"vof", buf, 64)),
        ("interp_velocity with a short buffer", lambda api, h, e: api.interp_velocity(h, buf, 64)),
        ("solve_p with an unknown criterion", lambda api, h, e: api.solve_p(h, 1e-6, 10, 5, 9, C.byref(i32), C.byref(out))),
        ("solve_p with max_iters < 0", lambda api, h, e: api.solve_p(h, 1e-6, -1, 5, 0, C.byref(i32), C.byref(out))),
        ("copy_rows from itself beyond its rows", lambda api, h, e: api.copy_rows(h, h, b"F", 0, e.nx + 7)),
        ("copy_rows of an unknown field", lambda api, h, e: api.copy_rows(h, h, b"vorticity", 0, 1)),
        ("a NULL handle", lambda api, h, e: api.step(None, 1)),
    ]


ABUSES = _abuses()


def draw_case(seed, large=0.03, huge=0.0):
    """Everything a case is made of, from its seed: a dict the replay needs nothing else for.  large: share of grids of 0.3-2 M
    cells (several chunk rows and tile columns of the pair kernels, chains of launches; the oracle needs seconds for those)."""
    rng = np.random.default_rng(seed)
    dtype = "f64" if rng.random() < 0.7 else "f32"
    shape = rng.choice(["tiny", "small", "square", "wide", "tall", "medium"], p=[0.1, 0.2, 0.25, 0.15, 0.1, 0.2])
    if rng.random() < huge:   # (what the rule of vof_step gives the pair kernels / the chains by itself: from 4 M / 6 M cells, nx >= 2048)
        shape = "large"
        nx, ny = int(rng.integers(2048, 3100)), int(rng.integers(1960, 3100))
        if rng.random() < 0.3:
            ny = nx
    elif rng.random() < large:
        shape = "large"
        nx, ny = int(rng.integers(520, 1500)), int(rng.integers(520, 1400))
        if rng.random() < 0.5:
            ny = nx
    elif shape == "tiny":
        nx, ny = int(rng.integers(3, 20)), int(rng.integers(3, 20))
    elif shape == "small":
        nx, ny = int(rng.integers(16, 140)), int(rng.integers(8, 140))
    elif shape == "square":
        nx = ny = int(rng.integers(16, 420))
    elif shape == "wide":
        nx, ny = int(rng.integers(16, 90)), int(rng.integers(200, 1200))
    elif shape == "tall":
        nx, ny = int(rng.integers(300, 900)), int(rng.integers(8, 130))
    else:
        nx, ny = int(rng.integers(100, 520)), int(rng.integers(100, 700))
    kw = {}
    cells = rng.choice(["default", "square", "free"], p=[0.4, 0.4, 0.2])
    if cells == "square" and nx != ny:
        kw["Lx"], kw["Ly"] = 0.1, 0.1 * ny / nx       # (dx == dy up to the rounding of the constants: the library decides)
    elif cells == "free":
        kw["Lx"], kw["Ly"] = float(rng.uniform(0.05, 0.2)), float(rng.uniform(0.05, 0.2))
    if rng.random() < 0.3:
        kw["sigma"] = float(rng.choice([0.0, 0.05, 0.0007, -0.001]))
    if rng.random() < 0.3:
        kw["gy"] = float(rng.choice([0.0, -0.0, -9.81, 3.0]))
    if rng.random() < 0.2:
        kw["gx"] = float(rng.choice([-0.0, 2.0, -4.0]))
    if rng.random() < 0.25:
        kw["dt"] = float(rng.choice([1e-6, 2e-6, 8e-6]))
    iters = int(rng.choice([10, 10, 10, 10, 10, 20, 20, 30, 2, 4, 6, 12, 5, 15, 1, 3, 7, 0]))   # (odd counts blow up within tens of steps)
    knobs = {}
    for name, vals in KNOBS.items():
        if rng.random() < 0.45:
            knobs[name] = int(rng.choice(vals))
    ops, budget = [], int(200000 * 60 / max(nx * ny, 1))      # cell-updates the oracle gets per case, in steps
    budget = max(44 if shape == "large" else 6, min(budget, 140))
    nops = int(rng.integers(3, 9))
    for _ in range(nops):
        kind = rng.choice(["step", "step", "step", "bigstep", "verbs", "reader", "phases", "set", "sigma", "knob", "sweeps", "solve", "tiny_p", "profile", "istep", "setrows", "abuse", "norms"])
        if kind == "step":
            ops.append(("step", int(rng.integers(1, 13))))
        elif kind == "bigstep":
            ops.append(("step", int(rng.choice([16, 19, 32, 33, 40, 41, 64, 75]))))
        elif kind == "verbs":
            ops.append(("verbs", int(rng.integers(1, 3)), int(rng.choice([iters if iters else 3, 3, 10]))))
        elif kind == "reader":
            ops.append(("reader", str(rng.choice(["cal_nu_rho", "get_normal_young", "advect_upwind", "solve_p_jacobi", "update_uv",
                                                  "fct_x_sweep", "fct_y_sweep", "post_process_f", "set_BC", "vis", "interp", "rows"]))))
        elif kind == "phases":
            ops.append(("phases",))
        elif kind == "set":
            ops.append(("set", str(rng.choice(["F", "u", "v", "p"])), int(rng.integers(0, 1 << 30))))
        elif kind == "sigma":
            ops.append(("sigma", float(rng.choice([0.0, 0.01, 0.007, 0.05]))))
        elif kind == "knob":
            name = str(rng.choice(list(KNOBS)))
            ops.append(("knob", name, int(rng.choice(KNOBS[name]))))
        elif kind == "sweeps":
            ops.append(("sweeps", int(rng.choice([1, 2, 5, 10, 13, 20, 25]))))
        elif kind == "solve":
            ops.append(("solve", int(rng.choice([10, 25, 40])), int(rng.choice([3, 5, 10])), str(rng.choice(["abs", "rel"]))))
        elif kind == "profile":
            ops.append(("profile", int(rng.choice([1, 2, 5, 18, 35]))))
        elif kind == "istep":
            ops.append(("istep", int(rng.choice([0, 1, 2, 7, 100, 1001]))))
        elif kind == "abuse":
            ops.append(("abuse", int(rng.integers(0, len(ABUSES)))))
        elif kind == "norms":
            ops.append(("norms", int(rng.choice([1, 2, 3, 5, 6, 10, 11, 17, 20])), int(rng.choice([0, 0, 1, 4, 5, 10, 13]))))
        elif kind == "setrows":
            ops.append(("setrows", str(rng.choice(["F", "u", "v", "p"])), float(rng.random()), float(rng.random()), int(rng.integers(0, 1 << 30))))
        else:
            ops.append(("tiny_p", int(rng.integers(0, 1 << 30))))
    # keep the oracle's work bounded: scale the step counts down to the budget
    total = sum(o[1] for o in ops if o[0] == "step")
    if total > budget:
        ops = [("step", max(1, o[1] * budget // total)) if o[0] == "step" else o for o in ops]
    return dict(seed=seed, nx=nx, ny=ny, dtype=dtype, cast=str(rng.choice(["f32", "f32", "none"])), ic=int(rng.integers(1, 4)),
                kw=kw, iters=iters, knobs=knobs, ops=ops, eager=bool(rng.random() < 0.08))


def describe(case):
    return "seed %d: %dx%d %s cast=%s ic=%d iters=%d%s kw=%r knobs=%r ops=%r" % (
        case["seed"], case["nx"], case["ny"], case["dtype"], case["cast"], case["ic"], case["iters"], " eager" if case.get("eager") else "",
        case["kw"], case["knobs"], case["ops"])


VERBS = ("cal_nu_rho", "get_normal_young", "advect_upwind", "set_BC", "solve_p_jacobi", "update_uv", "set_BC",
         "solve_VOF_rudman", "post_process_f", "set_BC")
PREFIX = {"advect_upwind": ("cal_nu_rho", "get_normal_young"), "solve_p_jacobi": ("cal_nu_rho",), "update_uv": ("cal_nu_rho",)}
EXTRA = {"cal_nu_rho": ("rho", "nu"), "get_normal_young": ("mx", "my", "kappa"),
         "advect_upwind": ("u_star", "v_star", "rho", "nu", "mx", "my", "kappa"), "solve_p_jacobi": ("rho", "nu"), "update_uv": ("rho", "nu")}


def all_finite(e, names=STATE):
    return all(bool(np.isfinite(e.get(n)).all()) for n in names)


def fields_differ(a, b, names):
    msgs = []
    for n in names:
        x, y = a.get(n), b.get(n)
        if not np.array_equal(x, y, equal_nan=True):
            msgs.append(diff_report(x, y, n))
    return msgs


def run_case(hip_api, oracle_api, case):
    """Replays a case on both engines; returns None or the description of the first divergence."""
    from vof2d.engine import VofError
    nx, ny, dtype = case["nx"], case["ny"], case["dtype"]
    try:
        a = engine(hip_api, nx, ny, dtype, case["cast"], ic=case["ic"], jacobi_iters=case["iters"],
                   flags=1 if case.get("eager") else 0, **case["kw"])     # (VOF_FLAG_NO_GRAPH: every launch eager)
    except VofError:
        try:   # (a constant the exact division cannot take: both sides must refuse)
            engine(oracle_api, nx, ny, dtype, case["cast"], ic=case["ic"], jacobi_iters=case["iters"], **case["kw"])
        except VofError:
            STATS["refused"] += 1
            return None
        return "the library refused a description the oracle accepts"
    b = engine(oracle_api, nx, ny, dtype, case["cast"], ic=case["ic"], jacobi_iters=case["iters"], **case["kw"])
    try:
        for k, v in case["knobs"].items():
            a.set_param(k, v)
        msgs = fields_differ(a, b, ("F",))
        if msgs:
            return "after set_init_F: " + " ; ".join(msgs)
        for n, op in enumerate(case["ops"]):
            names = STATE
            if op[0] == "step":
                a.step(op[1]); b.step(op[1])
                names = STATE + SCRATCH
            elif op[0] == "verbs":
                for _ in range(op[1]):
                    for e in (a, b):
                        istep = e.istep + 1
                        for verb in VERBS:
                            if verb == "solve_p_jacobi":
                                e.solve_p_jacobi(op[2])
                            elif verb == "solve_VOF_rudman":
                                e.solve_VOF_rudman(istep)
                            else:
                                getattr(e, verb)()
                        e.istep = istep
                names = STATE + ("u_star", "v_star", "mx", "my", "kappa", "rho", "nu")
            elif op[0] == "reader":
                r = op[1]
                if r == "vis":
                    for w in ("vof", "u", "v", "vnorm"):
                        if not np.array_equal(a.vis_field(w), b.vis_field(w), equal_nan=True):
                            return "op %d %r: vis field %s differs" % (n, op, w)
                elif r == "interp":
                    if not np.array_equal(a.interp_velocity(), b.interp_velocity(), equal_nan=True):
                        return "op %d %r: interp_velocity differs" % (n, op)
                elif r == "rows":
                    lo, hi = max(0, nx // 3 - 1), min(nx + 1, nx // 3 + 2)
                    for f in STATE:
                        if not np.array_equal(a.get(f, (lo, hi)), b.get(f, (lo, hi)), equal_nan=True):
                            return "op %d %r: rows %d..%d of %s differ" % (n, op, lo, hi, f)
                else:
                    for e in (a, b):
                        for verb in PREFIX.get(r, ()) + (r,):
                            if verb == "solve_p_jacobi":
                                e.solve_p_jacobi(3)
                            else:
                                getattr(e, verb)()
                    names = STATE + EXTRA.get(r, ())
            elif op[0] == "phases":
                for ph in (0, 1, 2):
                    a.step_phase(ph)
                b.step(1)
            elif op[0] == "profile":     # the in-situ profiler replays the handle's launch sequence eagerly: the state advances all the same
                if a.api.prefix == "vof_":
                    a.profile_steps(op[1])
                else:
                    a.step(op[1])
                b.step(op[1])
            elif op[0] == "set":
                rng = np.random.default_rng(op[2])
                f = op[1]
                x = b.get(f).astype(np.float64)
                if f == "F":
                    x = np.clip(x + 0.3 * rng.standard_normal(x.shape) * (rng.random(x.shape) < 0.1), 0, 1)
                elif f == "p":
                    x = x + rng.standard_normal(x.shape) * 10.0 ** rng.integers(-3, 3)
                else:
                    x = x + 0.02 * rng.standard_normal(x.shape)
                for e in (a, b):
                    e.set(f, x)
            elif op[0] == "sigma":
                for e in (a, b):
                    e.set_param("sigma", op[1])
            elif op[0] == "norms":       # sweeps whose last one reduces max|p_new - p| and max|p_new| (lane maxima -> __shfl_down -> atomicMax)
                # (op[1] sweeps on a rhs built now, then op[2] more on the same rhs -- the two calls of a residual-terminated solve;
                #  sweeps without a build on anything else are outside the contract: the oracle re-forms rhs from its rho array)
                for k, (cnt, build) in enumerate(((op[1], True), (op[2], False))):
                    if cnt == 0:
                        continue
                    ra, rb = a.jacobi_sweeps_norms(cnt, build), b.jacobi_sweeps_norms(cnt, build)
                    if ra != rb and not (np.isnan(ra).any() or np.isnan(rb).any()):
                        return "op %d %r, call %d: norms %r against %r" % (n, op, k, ra, rb)
                names = STATE      # (the library forms rho in registers for this rhs: the rho / nu arrays belong to the verbs)
            elif op[0] == "abuse":       # an invalid call: refused with a status, the state as it was
                what, call = ABUSES[op[1]]
                for e in (a, b):
                    rc = call(e.api, e.handle, e)
                    if rc == 0:
                        return "op %d: %s was accepted by %s" % (n, what, e.api.prefix)
            elif op[0] == "istep":       # (the step parity moved alone: the other sweep order, the other mask set of the work plan)
                for e in (a, b):
                    e.istep = op[1]
            elif op[0] == "setrows":     # a band of rows overwritten (vof_set_rows)
                rng = np.random.default_rng(op[4])
                g0 = int(op[2] * (nx + 1))
                g1 = min(nx + 1, g0 + int(op[3] * 40))
                x = b.get(op[1], (g0, g1)).astype(np.float64)
                x = np.clip(x + 0.2 * (rng.random(x.shape) < 0.2), 0, 1) if op[1] == "F" else x + 0.01 * rng.standard_normal(x.shape)
                for e in (a, b):
                    e.set(op[1], x, (g0, g1))
            elif op[0] == "knob":
                a.set_param(op[1], op[2])
            elif op[0] == "sweeps":
                for e in (a, b):
                    e.cal_nu_rho()
                    e.solve_p_jacobi(op[1])
                names = STATE + ("rho", "nu")
            elif op[0] == "solve":
                ra = a.solve_p(1e-30, op[1], op[2], op[3])
                rb = b.solve_p(1e-30, op[1], op[2], op[3])
                if ra != rb:
                    return "op %d %r: solve_p returned %r against %r" % (n, op, ra, rb)
            elif op[0] == "tiny_p":
                # a ring of tiny pressure values (the scaled tier of the exact division, the work plan of the Jacobi kernels)
                rng = np.random.default_rng(op[1])
                tiny = 1e-290 if dtype == "f64" else 1e-32
                i, j = np.meshgrid(np.arange(nx + 2), np.arange(ny + 2), indexing="ij")
                r = np.hypot(i - rng.uniform(0.3, 0.7) * nx, j - rng.uniform(0.3, 0.7) * ny)
                x = b.get("p").astype(np.float64)
                band = (r > 0.15 * min(nx, ny)) & (r < 0.4 * min(nx, ny))
                x[band] = tiny * rng.uniform(0.01, 50.0, size=int(band.sum())) * rng.choice([-1.0, 1.0], size=int(band.sum()))
                x[r <= 0.15 * min(nx, ny)] = 0.0
                for e in (a, b):
                    e.set("p", x)
            if not all_finite(b):
                # The run blew up (an odd sweep count leaves the checkerboard mode of the Jacobi iteration in p, twice the
                # time step ...): from the first inf / NaN on the two sides may part -- max / min are v_max / v_min in the
                # kernels and comparisons in the oracle, equal for every non-NaN pair only -- and a Courant
                # count taken while the NaN regions grew differently stays different.  Nothing after this point says anything.
                STATS["blew_up"] += 1
                return None
            if case["iters"] == 0:
                names = tuple(x for x in names if x != "rhs")   # (no sweep, no rhs: the reference builds it inside solve_p_jacobi)
            msgs = fields_differ(a, b, names)
            if msgs:
                return "op %d %r: " % (n, op) + " ; ".join(msgs)
            if a.istep != b.istep:
                return "op %d %r: istep %d against %d" % (n, op, a.istep, b.istep)
            ca, cb = a.get_counter("courant_violations"), b.get_counter("courant_violations")
            if ca != cb:
                return "op %d %r: courant_violations %d against %d" % (n, op, ca, cb)
            STATS["ops"] += 1
        STATS["completed"] += 1
        return None
    finally:
        if a.api.prefix == "vof_":
            for c in COVERAGE:
                try:
                    STATS[c] = STATS.get(c, 0) + (1 if a.get_counter(c) > 0 else 0)
                except VofError:
                    pass
        a.close(); b.close()


@pytest.mark.gpu
def test_random_call_sequences_match_the_oracle(hip_api, oracle_api):
    failures = []
    for k in range(NCASES):
        case = draw_case(SEED0 + k)
        why = run_case(hip_api, oracle_api, case)
        if why:
            failures.append(describe(case) + "\n    -> " + why)
    assert not failures, "%d of %d cases diverge:\n" % (len(failures), NCASES) + "\n".join(failures)


# ---------------------------------------------------------------------------------------------------- strips
def draw_strip_case(seed):
    """Row strips on one device (device copies stand in for the send / recv groups) against the single domain."""
    rng = np.random.default_rng(seed)
    dtype = "f64" if rng.random() < 0.65 else "f32"
    iters = int(rng.choice([10, 10, 10, 20, 5, 15, 30]))
    W = iters + 8                                   # VOF_HALO_ROWS (include/vof2d.h)
    nstrips = int(rng.integers(2, 7))
    nx = int(rng.integers(nstrips * (W + 1), max(nstrips * (W + 1) + 1, 900)))
    ny = int(rng.choice([rng.integers(8, 140), rng.integers(100, 700), rng.integers(600, 1300)]))
    if rng.random() < 0.5:
        ny = nx if nx >= 8 else ny
    # an uneven partition: every strip at least W rows
    extra = nx - nstrips * W
    cuts = np.sort(rng.integers(0, extra + 1, size=nstrips - 1))
    sizes = np.diff(np.concatenate([[0], cuts, [extra]])) + W
    bounds = np.concatenate([[0], np.cumsum(sizes)])
    owns = [(int(bounds[k]) + 1, int(bounds[k + 1])) for k in range(nstrips)]
    mode = str(rng.choice(["whole", "phased", "pieces", "pieces"]))
    kw = {}
    if nx != ny and rng.random() < 0.6:
        kw["Lx"], kw["Ly"] = 0.1, 0.1 * ny / nx
    if rng.random() < 0.2:
        kw["gy"] = float(rng.choice([0.0, -9.81]))
    knobs = {}
    for name in ("jacobi_pair", "buffer_stores", "tm_rows", "jacobi_pair_rows", "jacobi_tb_rows", "momentum_rows", "fctx_corr_rows",
                 "jacobi_tb_adapt", "virtual_ghosts", "pair_slow10"):
        if rng.random() < 0.35:
            knobs[name] = int(rng.choice(KNOBS[name]))
    budget = max(8, min(int(3.0e7 / (nx * ny)), 70))
    calls = [int(c) for c in rng.integers(1, 17, size=int(rng.integers(2, 7)))]
    while sum(calls) > budget and len(calls) > 1:
        calls.pop()
    return dict(seed=seed, nx=nx, ny=ny, dtype=dtype, ic=int(rng.integers(1, 4)), iters=iters, owns=owns, mode=mode, kw=kw, knobs=knobs,
                calls=calls, tiny=bool(rng.random() < 0.3), shallow=bool(rng.random() < 0.7))


def describe_strip(case):
    return "seed %d: %dx%d %s ic=%d iters=%d strips=%r mode=%s%s kw=%r knobs=%r calls=%r tiny=%r" % (
        case["seed"], case["nx"], case["ny"], case["dtype"], case["ic"], case["iters"], case["owns"], case["mode"],
        " (shallow halos of F, u*, v*)" if case["mode"] == "pieces" and case.get("shallow") else "", case["kw"],
        case["knobs"], case["calls"], case["tiny"])


def run_strip_case(hip_api, case):
    from vof2d.engine import VofError
    from vof2d.strips import stored_rows
    nx, ny, dtype, iters, owns = case["nx"], case["ny"], case["dtype"], case["iters"], case["owns"]
    W, n = iters + 8, len(case["owns"])
    mk = lambda **k: engine(hip_api, nx, ny, dtype, "f32", ic=case["ic"], jacobi_iters=iters, **dict(case["kw"], **k))
    full = mk()
    strips = [mk(rows=stored_rows(nx, o, W), own=o) for o in owns]
    try:
        for s in strips:
            for k, v in case["knobs"].items():
                s.set_param(k, v)
        if case["tiny"]:   # a ring of tiny pressure values across the strips (the work plan of the Jacobi kernels, strip by strip)
            rng = np.random.default_rng(case["seed"])
            i, j = np.meshgrid(np.arange(nx + 2), np.arange(ny + 2), indexing="ij")
            r = np.hypot(i - 0.5 * nx, j - 0.45 * ny)
            x = np.zeros((nx + 2, ny + 2))
            band = (r > 0.15 * min(nx, ny)) & (r < 0.45 * min(nx, ny))
            x[band] = (1e-290 if dtype == "f64" else 1e-32) * rng.uniform(0.01, 50.0, size=int(band.sum()))
            x[r <= 0.15 * min(nx, ny)] = 1.0
            full.set("p", x)
            for s in strips:
                s.set("p", x[s.row_lo:s.row_hi + 1])

        def trade(fields, D=W):
            for k in range(n - 1):
                lo_s, hi_s = strips[k], strips[k + 1]
                edge = owns[k][1]
                for f in fields:
                    lo_s.copy_rows_from(hi_s, f, edge + 1, edge + D)
                    hi_s.copy_rows_from(lo_s, f, edge + 1 - D, edge)

        def check(ctx):
            for k, s in enumerate(strips):
                g0 = 0 if k == 0 else owns[k][0]
                g1 = nx + 1 if k == n - 1 else owns[k][1]
                for f in STATE:
                    x, y = s.get(f, (g0, g1)), full.get(f, (g0, g1))
                    if not np.array_equal(x, y, equal_nan=True):
                        return "%s, strip %d (rows %d..%d): %s" % (ctx, k, owns[k][0], owns[k][1], diff_report(x, y, f))
            return None

        mode = case["mode"]
        if mode == "pieces":
            try:
                strips[0].step(1)
            except VofError:
                return None
            full.step(1)
            for s in strips[1:]:
                s.step(1)
            trade(STATE)
        done = 1 if mode == "pieces" else 0
        for call in case["calls"]:
            full.step(call)
            if mode == "whole":
                for _ in range(call):
                    for s in strips:
                        s.step(1)
                    trade(STATE)
            elif mode == "phased":
                for _ in range(call):
                    for ph, fields in ((0, ("p",)), (1, ("u", "v")), (2, ("F",))):
                        for s in strips:
                            s.step_phase(ph)
                        trade(fields)
            else:
                try:
                    for s in strips:
                        s.step_tm_piece(0)
                except VofError as e:      # (a knob combination the pair kernels do not take: the library says so)
                    if "pair kernels need" in str(e):
                        return None
                    raise
                trade(("u_star", "v_star", "rhs"))
                for _ in range(call - 1):
                    for s in strips:
                        s.step_tm_piece(1)
                    # what the library's own exchange ships in a middle step (runtime/comm.h, kTmReachRows): p and rhs W rows deep,
                    # F, u*, v* only the 8 rows the marches of the next k_tm read beyond the owned rows -- the deeper halo rows of
                    # those three keep whatever an earlier exchange left there
                    if case.get("shallow", True):
                        trade(("rhs", "p"))
                        trade(("F", "u_star", "v_star"), 8)
                    else:
                        trade(("F", "u_star", "v_star", "rhs", "p"))
                for s in strips:
                    s.step_tm_piece(2)
                trade(STATE)
            done += call
            if not all_finite(full):
                STATS["blew_up"] += 1
                return None
            why = check("after step %d (%s)" % (done, mode))
            if why:
                return why
            STATS["ops"] += 1
        ca, cb = sum(s.get_counter("courant_violations") for s in strips), full.get_counter("courant_violations")
        if ca != cb:
            return "courant_violations: strips %d, single domain %d" % (ca, cb)
        STATS["completed"] += 1
        STATS["strip_" + mode] = STATS.get("strip_" + mode, 0) + 1
        return None
    finally:
        full.close()
        for s in strips:
            s.close()


@pytest.mark.gpu
def test_random_strip_decompositions_match_the_single_domain(hip_api):
    failures = []
    for k in range(max(1, NCASES // 2)):
        case = draw_strip_case(SEED0 + k)
        why = run_strip_case(hip_api, case)
        if why:
            failures.append(describe_strip(case) + "\n    -> " + why)
    assert not failures, "%d cases diverge:\n" % len(failures) + "\n".join(failures)


@pytest.mark.gpu
def test_handles_on_concurrent_host_threads(hip_api):
    """One handle per host thread (one host thread per handle; ctypes releases the GIL inside a call): four threads
    create, step (graph captures, batch forms, chains), overwrite and read their own handles at the same time, several rounds;
    every thread's fields equal those of the same sequence run alone afterwards."""
    import threading
    specs = [(448, 400, "f64", 1, {"fuse_tm": 1, "overlap_halves": 0}), (300, 333, "f32", 2, {"fuse_tm": 1, "jacobi_pair": 2}),
             (700, 260, "f64", 3, {"fuse_tm": 0, "overlap_halves": 2, "batch_steps": 8}), (96, 130, "f64", 2, {})]

    def run(spec, out, k):
        nx, ny, dtype, ic, knobs = spec
        try:
            e = engine(hip_api, nx, ny, dtype, "f32", ic=ic)
            for name, v in knobs.items():
                e.set_param(name, v)
            for n in (1, 40, 7):
                e.step(n)
            u = e.get("u")
            u[3:9, 2:7] += 0.01
            e.set("u", u)
            e.step(33)
            out[k] = {f: e.get(f) for f in STATE}
            e.close()
        except Exception as exc:      # (reported by the main thread)
            out[k] = exc

    for _ in range(3):
        got = [None] * len(specs)
        threads = [threading.Thread(target=run, args=(sp, got, k)) for k, sp in enumerate(specs)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for k, sp in enumerate(specs):
            alone = [None]
            run(sp, alone, 0)
            assert not isinstance(got[k], Exception), (sp, got[k])
            assert not isinstance(alone[0], Exception), (sp, alone[0])
            for f in STATE:
                assert np.array_equal(got[k][f], alone[0][f], equal_nan=True), (sp, f)


@pytest.mark.gpu
def test_create_step_destroy_leaves_no_device_memory_behind(hip_api):
    """Handles come and go (every batch form captured, chains with their extra streams and events, a strip with a looped-back
    communicator's worth of graphs is test_host_gpu's business): after 60 create / step / destroy cycles the device has the memory
    it had after the first few, and the last handle still equals the first one's results."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(), C.c_size_t()

    def free_bytes():
        assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return free.value

    def cycle(k):
        nx, ny = (448, 400) if k % 3 else (700, 260)
        e = engine(hip_api, nx, ny, "f64" if k % 2 else "f32", "f32", ic=1 + k % 3)
        e.set_param("fuse_tm", 1 if k % 3 else 0)
        e.set_param("overlap_halves", 0 if k % 3 else 2)
        e.set_param("batch_steps", 8)
        e.step(41)
        out = e.get("F")
        e.close()
        return out

    first = [cycle(k) for k in range(6)]
    base = free_bytes()
    for k in range(6, 60):
        last = cycle(k)
        if k >= 54:
            assert np.array_equal(last, first[k - 54]), k       # (k and k - 54 draw the same case)
    assert base - free_bytes() < (64 << 20), (base, free_bytes())


@pytest.mark.gpu
def test_handles_on_a_caller_stream_and_two_handles_on_one_stream(hip_api):
    """vof_create with the caller's stream (the library then creates none of its own for the step; the chains' extra streams only
    ever run inside graphs launched on it): two handles sharing ONE caller stream, stepped alternately through every batch form, equal
    the same handles on streams of their own."""
    import ctypes as C
    from vof2d.engine import Engine, make_desc
    hip = C.CDLL("libamdhip64.so")
    st = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(st), 1) == 0      # hipStreamNonBlocking
    specs = [(448, 400, "f64", 1, {"fuse_tm": 1, "overlap_halves": 0}), (700, 260, "f32", 3, {"fuse_tm": 0, "overlap_halves": 2, "batch_steps": 8})]
    shared, own = [], []
    for nx, ny, dtype, ic, knobs in specs:
        for group, stream in ((shared, st.value), (own, None)):
            e = Engine(hip_api, make_desc(hip_api, nx, ny, dtype, "f32"), stream=stream)
            e.set_init_F(ic)
            for k, v in knobs.items():
                e.set_param(k, v)
            group.append(e)
    for n in (1, 40, 3, 18, 33):
        for e in shared + own:
            e.step(n)
    for a, b in zip(shared, own):
        for f in STATE + SCRATCH:
            assert np.array_equal(a.get(f), b.get(f), equal_nan=True), f
        assert a.get_counter("courant_violations") == b.get_counter("courant_violations")
    for e in shared + own:
        e.close()
    assert hip.hipStreamDestroy(st) == 0


def _every_abuse_is_refused(api):
    for nx, ny in ((20, 24), (300, 260)):
        e = engine(api, nx, ny, "f64", "f32", ic=1)
        e.step(2)
        before = {f: e.get(f) for f in STATE}
        for what, call in ABUSES:
            assert call(api, e.handle, e) != 0, what
        for f in STATE:
            assert np.array_equal(before[f], e.get(f)), f
        e.step(1)
        e.close()


@pytest.mark.gpu
def test_invalid_calls_are_refused_and_change_nothing(hip_api):
    """Every entry of ABUSES straight through the ABI (NULL pointers, rows outside the strip, short buffers, unknown names,
    phases out of order ...): a status code each, no crash, the fields as they were, the next step runs."""
    _every_abuse_is_refused(hip_api)


def test_invalid_calls_are_refused_by_the_oracle_too(oracle_api):
    _every_abuse_is_refused(oracle_api)


def test_the_generator_is_deterministic_and_the_replay_runs_on_the_oracle(oracle_api):
    """CPU leg: the same seed draws the same case, and the replay itself (oracle against oracle, no knobs) goes through
    every kind of call without a divergence -- so a failure of the GPU leg is the library's."""
    assert draw_case(SEED0) == draw_case(SEED0)
    kinds = set()
    for k in range(400):
        kinds.update(o[0] for o in draw_case(SEED0 + k)["ops"])
    assert kinds == {"step", "verbs", "reader", "phases", "set", "sigma", "knob", "sweeps", "solve", "tiny_p", "profile", "istep", "setrows", "abuse", "norms"}
    done = 0
    for k in range(60):
        case = draw_case(SEED0 + k)
        if case["nx"] * case["ny"] > 6000:
            continue
        case = dict(case, knobs={}, ops=[o for o in case["ops"] if o[0] not in ("knob", "phases")])
        assert run_case(oracle_api, oracle_api, case) is None, describe(case)
        done += 1
    assert done >= 5


if __name__ == "__main__":
    import argparse
    import ctypes
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=SEED0)
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--log", default=None)
    ap.add_argument("--seconds", type=float, default=0.0, help="stop after this much wall time (0: run all cases)")
    ap.add_argument("--large", type=float, default=0.03, help="share of grids of 0.3-2 M cells")
    ap.add_argument("--huge", type=float, default=0.0, help="share of grids of 4-9 M cells with nx >= 2048 (the sizes the rule of vof_step acts on)")
    ap.add_argument("--strips", action="store_true", help="the strip cases (library against library) instead of the call sequences")
    args = ap.parse_args()
    from vof2d import _abi
    from vof2d._lib import hip_api as load
    os.environ.setdefault("OMP_NUM_THREADS", "16")
    oracle = _abi.bind(ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libvof_oracle.so")), "ovof_", optional=_abi.GPU_ONLY)
    hip = load()
    out = open(args.log, "w") if args.log else sys.stdout
    t0, bad, ran = time.time(), 0, 0
    for k in range(args.cases):
        if args.seconds and time.time() - t0 > args.seconds:
            break
        if args.strips:
            case = draw_strip_case(args.seed + k)
            why, text = run_strip_case(hip, case), describe_strip(case)
        else:
            case = draw_case(args.seed + k, args.large, args.huge)
            why, text = run_case(hip, oracle, case), describe(case)
        ran += 1
        if why:
            bad += 1
            print("DIVERGES " + text + "\n    -> " + why, file=out, flush=True)
    print("fuzz: %d cases from seed %d, %d diverge, %.0f s; %d ran to their end, %d blew up on the way (compared up to there), %d refused by both sides; %d calls compared" % (
        ran, args.seed, bad, time.time() - t0, STATS["completed"], STATS["blew_up"], STATS["refused"], STATS["ops"]), file=out, flush=True)
    print("cases in which the library ran: " + ", ".join("%s %d" % (c, STATS.get(c, 0)) for c in COVERAGE) +
          "".join("; %s %d" % (k, v) for k, v in sorted(STATS.items()) if k.startswith("strip_")), file=out, flush=True)
    sys.exit(1 if bad else 0)
