"""The oracle against itself: the NumPy restatement and the scalar-C restatement of
2dvof.py are written independently and must agree value for value on every field
(SURVEY 8c "what pins the build's results" (1)), plus physics properties (3)."""
import numpy as np
import pytest

import vof_oracle_np as onp
from util import engine, same, diff_report

ALL = ("F", "u", "v", "p", "u_star", "v_star", "mx", "my", "kappa", "rho", "nu", "Ftd", "ax", "ay", "cx", "cy",
       "rp", "rm")
NPDT = {"f64": np.float64, "f32": np.float32}


def pair(oracle_api, nx, ny, ic, dtype, cast):
    s = onp.new_state(nx, ny, ic, dtype=NPDT[dtype], coord_cast=cast)
    e = engine(oracle_api, nx, ny, dtype, cast, ic=ic)
    return s, e


@pytest.mark.parametrize("nx,ny,ic,dtype,cast,steps", [
    (32, 32, 1, "f64", "f32", (1, 2, 3, 10, 100)),
    (33, 17, 2, "f64", "f32", (1, 2, 10, 60)),
    (24, 40, 3, "f64", "none", (1, 2, 10, 60)),
    (32, 32, 1, "f32", "f32", (1, 2, 10, 100)),
    (20, 48, 2, "f32", "f32", (1, 10, 60)),
    (16, 16, 3, "f32", "none", (1, 10, 60)),
])
def test_numpy_and_c_restatements_agree(oracle_api, nx, ny, ic, dtype, cast, steps):
    s, e = pair(oracle_api, nx, ny, ic, dtype, cast)
    assert same(e.get("F"), s.F), diff_report(e.get("F"), s.F, "init F")
    done = 0
    for st in steps:
        onp.step(s, st - done)
        e.step(st - done)
        done = st
        for f in ALL:
            a, b = e.get(f), getattr(s, f)
            assert same(a, b), "step %d %s" % (st, diff_report(a, b, f))
        assert e.get_counter("courant_violations") == s.courant_violations


def test_constants_fold_like_python(oracle_api):
    for nx, ny, dtype, cast in [(128, 128, "f64", "f32"), (200, 200, "f32", "f32"), (1024, 1024, "f64", "f32"),
                                (4096, 4096, "f64", "f32"), (8192, 8192, "f64", "none"), (100, 37, "f64", "none")]:
        p = onp.Params(nx, ny, dtype=NPDT[dtype], coord_cast=cast)
        e = engine(oracle_api, nx, ny, dtype, cast) if nx <= 1024 else None
        py = dict(dx=p.dx_d, dy=p.dy_d, dxi=p.dxi_d, dyi=p.dyi_d, dxi2=p.dxi_d ** 2, dyi2=p.dyi_d ** 2,
                  nrm_x=-1 / (2 * p.dx_d), kap_y=1 / p.dy_d / 2, dxdy=p.dx_d * p.dy_d, dtdy=p.dt_d * p.dy_d,
                  dtdx=p.dt_d * p.dx_d, cfl_x=0.25 * p.dx_d, half_dy=p.dy_d / 2, sqrt2dx=2.0 ** 0.5 * p.dx_d)
        assert p.dxi_d ** 2 == p.dxi_d * p.dxi_d
        if e is not None:
            for k, v in py.items():
                assert e.get_param(k) == v, (nx, k)
    # SURVEY 8c-S2 quoted values
    assert onp.Params(128, 128).dx_d == 0.00078125001164153218
    assert onp.Params(200, 200, dtype=np.float32).dx_d == 0.00050000002374872565
    assert onp.Params(4096, 4096).dx_d == 2.4414062863797881e-05


def test_dam_break_tie_row(oracle_api):
    """the row y[j] == Ly/2 is in for f32 / uncast f64 and out for f64 with the f32 cast."""
    for dtype, cast, rows in [("f32", "f32", 66), ("f64", "f32", 65), ("f64", "none", 66)]:
        F = engine(oracle_api, 128, 128, dtype, cast, ic=1).get("F")
        assert F.sum() == 44 * rows, (dtype, cast)
        assert set(np.unique(F)) == {0.0, 1.0}


@pytest.mark.parametrize("ic", [1, 2, 3])
def test_bounds_mass_and_symmetry(oracle_api, ic):
    e = engine(oracle_api, 64, 64, "f64", "f32", ic=ic)
    m0 = e.get("F")[1:-1, 1:-1].sum()
    e.step(200)
    F = e.get("F")
    assert F.min() >= 0.0 and F.max() <= 1.0
    assert abs(F[1:-1, 1:-1].sum() - m0) < 1e-3 * max(1.0, m0)
    if ic in (2, 3):  # left-right symmetric initial conditions stay symmetric to rounding
        assert np.max(np.abs(F[1:-1] - F[1:-1][::-1])) < 1e-6
    assert e.get_counter("courant_violations") == 0


def test_hydrostatic_pool_stays_nearly_at_rest(oracle_api):
    """A flat pool (F = 1 below mid-height) under gravity.  Ten Jacobi sweeps do not converge the
    Neumann problem, so small wall-driven velocities appear (the reference behaves the same way,
    2dvof.py:521); they stay tiny, mirror-symmetric in x, and the interface does not move."""
    e = engine(oracle_api, 32, 32, "f64", "none")
    F0 = np.zeros((34, 34))
    F0[:, :17] = 1.0
    e.set("F", F0)
    e.step(50)
    u, F = e.get("u"), e.get("F")
    assert np.max(np.abs(u)) < 1e-4
    assert np.max(np.abs(u[1:34] + u[1:34][::-1])) < 1e-12   # u[i] = -u[nx+2-i] (face-centred)
    assert np.max(np.abs(F - F[::-1])) < 1e-12
    assert np.max(np.abs(F - F0)) < 1e-4
    assert abs(F[1:-1, 1:-1].sum() - F0[1:-1, 1:-1].sum()) < 1e-3


def test_strip_decomposition_is_value_invariant(oracle_api):
    """Two strips with VOF_HALO_ROWS deep halos, exchanged once per step, reproduce the
    single-domain run exactly on their owned rows (DESIGN.md "strips")."""
    from vof2d import halo_rows
    nx, ny, W = 64, 24, halo_rows(10)
    full = engine(oracle_api, nx, ny, "f64", "f32", ic=1)
    mid = nx // 2
    a = engine(oracle_api, nx, ny, "f64", "f32", ic=1, rows=(0, mid + W), own=(1, mid))
    b = engine(oracle_api, nx, ny, "f64", "f32", ic=1, rows=(mid + 1 - W, nx + 1), own=(mid + 1, nx))
    for step in range(1, 41):
        full.step(1); a.step(1); b.step(1)
        for f in ("F", "u", "v", "p"):
            a.copy_rows_from(b, f, mid + 1, mid + W)
            b.copy_rows_from(a, f, mid + 1 - W, mid)
        for f in ("F", "u", "v", "p"):
            assert same(a.get(f, (0, mid)), full.get(f, (0, mid))), "step %d %s strip a" % (step, f)
            assert same(b.get(f, (mid + 1, nx + 1)), full.get(f, (mid + 1, nx + 1))), "step %d %s strip b" % (step, f)


def test_halo_narrower_than_required_breaks_invariance(oracle_api):
    """The deep halo is needed: 7 rows fewer (11 < the 15 the dependency analysis in DESIGN.md asks
    for) and the strips drift from the single-domain run."""
    from vof2d import halo_rows
    nx, ny, W = 64, 24, halo_rows(10) - 7
    full = engine(oracle_api, nx, ny, "f64", "f32", ic=1)
    mid = nx // 2
    a = engine(oracle_api, nx, ny, "f64", "f32", ic=1, rows=(0, mid + W), own=(1, mid))
    b = engine(oracle_api, nx, ny, "f64", "f32", ic=1, rows=(mid + 1 - W, nx + 1), own=(mid + 1, nx))
    ok = True
    for step in range(1, 41):
        full.step(1); a.step(1); b.step(1)
        for f in ("F", "u", "v", "p"):
            a.copy_rows_from(b, f, mid + 1, mid + W)
            b.copy_rows_from(a, f, mid + 1 - W, mid)
        ok = ok and same(a.get("p", (1, mid)), full.get("p", (1, mid)))
    assert not ok


def test_phased_step_equals_literal_main_loop(oracle_api):
    """ovof_step_phase (per-field boundary schedule, used under the overlapped halo exchange)
    leaves every field, ghosts included, exactly as the literal main loop of ovof_step does."""
    for ic, dtype in ((1, "f64"), (2, "f64"), (3, "f32")):
        a = engine(oracle_api, 40, 28, dtype, "f32", ic=ic)
        b = engine(oracle_api, 40, 28, dtype, "f32", ic=ic)
        for step in range(1, 16):
            a.step(1)
            for ph in (0, 1, 2):
                b.step_phase(ph)
            for f in ("F", "u", "v", "p", "u_star", "v_star"):
                assert same(a.get(f), b.get(f)), "ic %d step %d %s" % (ic, step, diff_report(a.get(f), b.get(f), f))
        assert a.istep == b.istep == 15


def test_display_fields_numpy_vs_c(oracle_api):
    """2dvof.py:458-492: get_*_field images and interp_velocity, both restatements."""
    s = onp.new_state(24, 18, 2, dtype=np.float64)
    e = engine(oracle_api, 24, 18, "f64", "f32", ic=2)
    onp.step(s, 30)
    e.step(30)
    for which in ("vof", "u", "v", "vnorm"):
        img = e.vis_field(which)
        assert img.shape == (48, 36)
        assert same(img, onp.vis_field(s, which)), which
    assert same(e.vis_field("vof")[:2, :2], np.full((2, 2), s.F[0, 0]))   # ghost entry 0, repeated 2 x 2
    V = e.interp_velocity()
    assert V.shape == (26, 20, 2) and same(V, onp.interp_velocity(s))
    assert not V[0].any() and not V[:, 0].any() and not V[:, 19].any()
