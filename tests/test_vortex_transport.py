"""Frozen single-vortex transport: the known-physics check of the FCT sweeps.

The reference's only "test" of fct_x_sweep / fct_y_sweep is test/forward_fct.py: F with a circular
hole is transported by a frozen analytic velocity field (the Kothe-Rider / LeVeque single vortex,
u = -sin^2(x) sin(2y) A, v = sin^2(y) sin(2x) A, test/forward_fct.py:196-204) through
solve_VOF_rudman and a human looks at the contour plot (no assertions).  Here the
same set-up runs through 2dvof.py's own sweeps (:312-448 + post_process_f :452-455 + set_BC
:162-189) with assertions instead of a picture:

  * 0 <= F <= 1 at every step; on square cells the mass sum(F) over the interior stays within the
    scheme's own splitting error (each sweep divides by its dv = dx dy - dt dy du, :324 / :388, so the
    pair conserves mass to second order in dt grad(u) only: a few 1e-5 relative over 100 steps.  On
    non-square cells the reference's y sweep scales its fluxes by dy/(dx dy) like the x sweep, :393,
    and is not conservative -- reproduced as is, compared value for value only),
  * reversing the velocity after n steps brings the hole back (time-reversed vortex),
  * the NumPy and C restatements agree value for value (CPU), and the HIP kernels equal the C
    oracle value for value (GPU), at every checked step.

The velocity comes from the stream function psi = A L / pi * sin^2(pi x / L) sin^2(pi y / L) sampled
at the cell corners and differenced along the faces, so it is discretely divergence free and its
wall-normal components vanish (what set_BC enforces anyway): u = -d(psi)/dy is the analytic
-sin^2(X) sin(2Y) A to second order.
"""
import numpy as np
import pytest

import vof_oracle_np as onp
from util import engine, same, diff_report, assert_fields_same


def vortex_velocity(nx, ny, dx, dy, amp, Lx=0.1, Ly=0.1):
    """(u, v) on the staggered (nx+2, ny+2) layout: u[i, j] on the left face of cell (i, j),
    v[i, j] on its bottom face (2dvof.py:179-185 / SURVEY 8a-1); ghost entries 0."""
    xi = (np.arange(nx + 3) - 1) * dx        # corner i sits at x = (i - 1) dx
    yj = (np.arange(ny + 3) - 1) * dy
    X, Y = np.meshgrid(np.pi * xi / Lx, np.pi * yj / Ly, indexing="ij")
    psi = amp * Lx / np.pi * np.sin(X) ** 2 * np.sin(Y) ** 2
    psi[(xi < 0) | (xi > Lx * (1 + 1e-12)), :] = 0.0
    psi[:, (yj < 0) | (yj > Ly * (1 + 1e-12))] = 0.0
    psi[1, :] = psi[nx + 1, :] = 0.0         # the walls are streamlines exactly
    psi[:, 1] = psi[:, ny + 1] = 0.0
    u = np.zeros((nx + 2, ny + 2))
    v = np.zeros((nx + 2, ny + 2))
    u[1:nx + 2, 1:ny + 1] = -(psi[1:nx + 2, 2:ny + 2] - psi[1:nx + 2, 1:ny + 1]) / dy
    v[1:nx + 1, 1:ny + 2] = (psi[2:nx + 2, 1:ny + 2] - psi[1:nx + 1, 1:ny + 2]) / dx
    return u, v


def transport(e, n, istep0):
    """n steps of the transport part of 2dvof.py's main loop (:526-528) with the velocity frozen."""
    for k in range(n):
        e.solve_VOF_rudman(istep0 + k + 1)
        e.post_process_f()
        e.set_BC()


def setup(api, nx, ny, dtype, sign=1.0, cfl=0.2):
    e = engine(api, nx, ny, dtype, "f32", ic=2)        # F = 1 with a circular hole (find_area, :148-152)
    dx, dy, dt = e.get_param("dx"), e.get_param("dy"), e.get_param("dt")
    u, v = vortex_velocity(nx, ny, dx, dy, cfl * dx / dt)
    T = np.float64 if dtype == "f64" else np.float32
    e.set("u", (sign * u).astype(T)); e.set("v", (sign * v).astype(T))
    e.set_BC()
    return e, u.astype(T), v.astype(T)


def interior(a):
    return a[1:-1, 1:-1].astype(np.float64)


def check_bounds_and_mass(F, mass0, tol):
    assert F.min() >= 0.0 and F.max() <= 1.0
    m = interior(F).sum()
    assert abs(m - mass0) <= tol * mass0, (m, mass0)


def test_velocity_is_discretely_divergence_free():
    nx, ny, dx = 40, 56, 0.1 / 40
    u, v = vortex_velocity(nx, ny, dx, 0.1 / 56, 3.0)
    div = (u[2:nx + 2, 1:ny + 1] - u[1:nx + 1, 1:ny + 1]) / dx + (v[1:nx + 1, 2:ny + 2] - v[1:nx + 1, 1:ny + 1]) / (0.1 / 56)
    assert np.abs(div).max() < 1e-9 * np.abs(u).max() / dx
    assert not u[1].any() and not u[nx + 1].any() and not v[:, 1].any() and not v[:, ny + 1].any()
    # second-order agreement with the analytic field of test/forward_fct.py:202-203 at a face centre
    i, j = 13, 20
    X, Y = np.pi * (i - 1) * dx / 0.1, np.pi * (j - 0.5) * (0.1 / 56) / 0.1
    assert abs(u[i, j] - (-np.sin(X) ** 2 * np.sin(2 * Y) * 3.0)) < 5e-3 * 3.0


MASS_TOL = 2e-4   # relative drift of sum(F) the direction-split scheme itself produces (see above)


@pytest.mark.parametrize("nx,ny,dtype", [(48, 48, "f64"), (48, 48, "f32"), (48, 40, "f64"), (30, 64, "f32")])
def test_vortex_oracles_agree_and_conserve(oracle_api, nx, ny, dtype):
    """CPU: C restatement == NumPy restatement through 60 vortex steps; bounds (and, on square
    cells, mass) hold."""
    e, u, v = setup(oracle_api, nx, ny, dtype)
    s = onp.new_state(nx, ny, 2, dtype=np.float64 if dtype == "f64" else np.float32, coord_cast="f32")
    s.u[...] = u; s.v[...] = v
    onp.set_BC(s)
    mass0 = interior(e.get("F")).sum()
    for blk in range(6):
        transport(e, 10, 10 * blk)
        for _ in range(10):
            s.istep += 1
            onp.solve_VOF_rudman(s); onp.post_process_f(s); onp.set_BC(s)
        F = e.get("F")
        assert same(F, s.F), "step %d %s" % (10 * blk + 10, diff_report(F, s.F, "F"))
        check_bounds_and_mass(F, mass0, MASS_TOL if nx == ny else 1.0)


def test_vortex_reversal_restores_the_hole(oracle_api):
    """Forward n steps, velocity reversed, n steps back: F returns to the initial hole up to the
    scheme's diffusion (L1 error a small fraction of the hole's area), and stays in bounds."""
    nx = ny = 64
    e, u, v = setup(oracle_api, nx, ny, "f64", cfl=0.25)
    F0 = e.get("F")
    hole = (1.0 - interior(F0)).sum()
    n = 120
    transport(e, n, 0)
    moved = np.abs(interior(e.get("F")) - interior(F0)).sum()
    assert moved > 0.2 * hole                      # the vortex really deformed the hole
    e.set("u", -u); e.set("v", -v); e.set_BC()
    transport(e, n, n)
    F1 = e.get("F")
    check_bounds_and_mass(F1, interior(F0).sum(), MASS_TOL)
    err = np.abs(interior(F1) - interior(F0)).sum()
    assert err < 0.2 * hole and err < 0.1 * moved, (err, moved, hole)


@pytest.mark.gpu
@pytest.mark.parametrize("nx,ny,dtype", [(96, 96, "f64"), (70, 133, "f64"), (128, 64, "f32")])
def test_vortex_gpu_equals_oracle(hip_api, oracle_api, nx, ny, dtype):
    a, _, _ = setup(hip_api, nx, ny, dtype)
    b, u, v = setup(oracle_api, nx, ny, dtype)
    assert_fields_same(a, b, ("F", "u", "v"), ctx="vortex set-up")
    mass0 = interior(b.get("F")).sum()
    for blk in range(5):
        transport(a, 16, 16 * blk); transport(b, 16, 16 * blk)
        assert_fields_same(a, b, ("F",), ctx="vortex step %d" % (16 * blk + 16))
        check_bounds_and_mass(a.get("F"), mass0, MASS_TOL if nx == ny else 1.0)
    for e in (a, b):                                # and back again
        e.set("u", -u); e.set("v", -v); e.set_BC()
        transport(e, 32, 80)
    assert_fields_same(a, b, ("F",), ctx="vortex reversed")


@pytest.mark.gpu
def test_vortex_gpu_large_grid_properties(hip_api):
    """2048^2 (no oracle at this size in seconds): bounds, mass, and the reversal property."""
    n = 2048
    e, u, v = setup(hip_api, n, n, "f64", cfl=0.25)
    F0 = e.get("F")
    mass0, hole = interior(F0).sum(), (1.0 - interior(F0)).sum()
    transport(e, 200, 0)
    check_bounds_and_mass(e.get("F"), mass0, 2e-5)     # the splitting error shrinks with dt grad(u)
    e.set("u", -u); e.set("v", -v); e.set_BC()
    transport(e, 200, 200)
    F1 = e.get("F")
    check_bounds_and_mass(F1, mass0, 2e-5)
    assert np.abs(interior(F1) - interior(F0)).sum() < 0.05 * hole
