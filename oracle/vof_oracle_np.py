"""
TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

NumPy restatement of the hot path of the reference solver
(/root/reference/2dvof.py), vectorised one statement group per Taichi
top-level ``for`` (every such loop is a barrier).

PARITY PIN: the reference needs ``taichi==1.4.1``, which is not installable in
this image (no wheel, no network), and its ``test/`` scripts hold no golden
vectors.  What this file is checked against instead is the reference's OWN
SOURCE TEXT, executed: ``tests/golden/make_ref_golden.py`` runs
``/root/reference/2dvof.py`` unmodified in the build container under a
pure-Python stand-in for the ``taichi`` module (200 x 200 as shipped, doubles,
-ic 1/2/3, 1000 steps; plus three rectangular-cell runs with only the grid-size
literals replaced) and commits the results as ``tests/golden/ref_*.npz``;
``tests/test_ref_golden.py`` requires this module (first 20 steps) and the C
restatement (every recorded step) to reproduce all 19 arrays of those runs
exactly.  NOT pinned: Taichi's code generation (``fast_math`` contraction /
reassociation) -- real Taichi output may differ from the reference's text in
the last bits.  It also agrees value-for-value with the independent scalar-C
restatement in ``oracle/vof_oracle.c`` (``tests/test_oracle.py``) and with the
committed self-generated fixtures under ``tests/golden/``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module.

Canonical arithmetic:
  * every field value has dtype T (float64 or float32);
  * expressions made only of Python-scope numbers in the reference
    (``dxi**2``, ``dx*dy``, ``dt*dy``, ``-1/(2*dx)``, ``1/dx/2``, ``Lx/3`` ...)
    are folded in Python double and then rounded once to T;
  * everything else is evaluated in T in Python precedence order,
    left to right, with no FMA contraction (NumPy ufuncs never contract);
  * max/min are the comparisons ``a if a > b else b`` / ``a if a < b else b``.
"""
import math

import numpy as np


def _vmax(a, b):
    # ti.max(a, b): (a > b) ? a : b
    return np.where(a > b, a, b)


def _vmin(a, b):
    # ti.min(a, b): (a < b) ? a : b
    return np.where(a < b, a, b)


def var(a, b, c):
    """2dvof.py:192-195 -- ``a + b + c - max(a,b,c) - min(a,b,c)`` left to right."""
    return ((a + b) + c) - _vmax(_vmax(a, b), c) - _vmin(_vmin(a, b), c)


class Params:
    """Constants of 2dvof.py:19-50 with the folding rule of S2/S9."""

    def __init__(self, nx, ny, dtype=np.float64, coord_cast="f32",
                 Lx=0.1, Ly=0.1, rho_l=1000.0, rho_g=50.0, nu_l=1.0e-6,
                 nu_g=1.5e-5, sigma=0.007, gx=0, gy=-5, dt=4e-6):
        T = np.dtype(dtype).type
        self.T = T
        self.nx, self.ny = int(nx), int(ny)
        self.Lx, self.Ly = Lx, Ly
        self.imin, self.jmin = 1, 1                      # :37-40
        self.imax, self.jmax = nx, ny
        # :43-46  xnp = hstack((0, linspace(0, Lx, nx+1), Lx)).astype(float32)
        xnp = np.hstack((0.0, np.linspace(0, Lx, nx + 1), Lx))
        ynp = np.hstack((0.0, np.linspace(0, Ly, ny + 1), Ly))
        if coord_cast == "f32":
            xnp = xnp.astype(np.float32)
            ynp = ynp.astype(np.float32)
        elif coord_cast != "none":
            raise ValueError("coord_cast must be 'f32' or 'none'")
        self.coord_cast = coord_cast
        # the x / y fields hold dtype-T values (x = ti.field(float, ...))
        self.x = xnp.astype(T)
        self.y = ynp.astype(T)
        # :47-50 Python-scope reads -> Python doubles
        dx = float(self.x[self.imin + 2]) - float(self.x[self.imin + 1])
        dy = float(self.y[self.jmin + 2]) - float(self.y[self.jmin + 1])
        dxi = 1 / dx
        dyi = 1 / dy
        self.dx_d, self.dy_d, self.dxi_d, self.dyi_d, self.dt_d = dx, dy, dxi, dyi, dt
        # rounded-once constants of type T
        self.dx, self.dy, self.dxi, self.dyi = T(dx), T(dy), T(dxi), T(dyi)
        self.dxi2, self.dyi2 = T(dxi ** 2), T(dyi ** 2)
        self.dt = T(dt)
        self.rho_l, self.rho_g = T(rho_l), T(rho_g)
        self.nu_l, self.nu_g = T(nu_l), T(nu_g)
        self.sigma = T(sigma)              # runtime 0-D field (:28-29)
        self.gx, self.gy = T(gx), T(gy)
        self.nrm_x = T(-1 / (2 * dx))      # :287  -1 / (2 * dx)
        self.nrm_y = T(-1 / (2 * dy))
        self.kap_x = T(1 / dx / 2)         # :308  1 / dx / 2
        self.kap_y = T(1 / dy / 2)
        self.dxdy = T(dx * dy)             # :324
        self.dtdy = T(dt * dy)             # :324  dt * dy
        self.dtdx = T(dt * dx)             # :388  dt * dx
        self.cfl_x = T(0.25 * dx)          # :274
        self.cfl_y = T(0.25 * dy)          # :279
        self.half_dx = T(dx / 2)           # :105
        self.half_dy = T(dy / 2)
        self.sqrt2dx = T(math.sqrt(2.0) * dx)  # :131
        self.tiny = T(1e-10)               # :300


FIELDS = ("F", "Ftd", "ax", "ay", "cx", "cy", "rp", "rm", "u", "v", "u_star",
          "v_star", "p", "pt", "rho", "nu", "mx", "my", "kappa")


class State:
    """The module-global ti.fields of 2dvof.py:53-89, zero-initialised (S5)."""

    def __init__(self, prm):
        self.prm = prm
        shp = (prm.nx + 2, prm.ny + 2)
        for name in FIELDS:
            setattr(self, name, np.zeros(shp, dtype=prm.T))
        self.istep = 0
        self.courant_violations = 0

    def copy_fields(self, names=("F", "u", "v", "p")):
        return {n: getattr(self, n).copy() for n in names}


# --------------------------------------------------------------------------
# 2dvof.py:102-134
def find_area(prm, cx, cy, r):
    T = prm.T
    nx, ny = prm.nx, prm.ny
    i = np.arange(nx + 2, dtype=np.int32)[:, None]
    j = np.arange(ny + 2, dtype=np.int32)[None, :]
    zero = np.zeros((nx + 2, ny + 2), dtype=T)
    xcoord_ct = (i - prm.imin).astype(T) * prm.dx + prm.half_dx + zero
    ycoord_ct = (j - prm.jmin).astype(T) * prm.dy + prm.half_dy + zero
    xcoord_lu = xcoord_ct - prm.half_dx
    ycoord_lu = ycoord_ct + prm.half_dy
    xcoord_ld = xcoord_ct - prm.half_dx
    ycoord_ld = ycoord_ct - prm.half_dy
    xcoord_ru = xcoord_ct + prm.half_dx
    ycoord_ru = ycoord_ct + prm.half_dy
    xcoord_rd = xcoord_ct + prm.half_dx
    ycoord_rd = ycoord_ct - prm.half_dy

    def dist(xc, yc):
        ddx = xc - cx
        ddy = yc - cy
        return np.sqrt(ddx * ddx + ddy * ddy)

    dist_ct = dist(xcoord_ct, ycoord_ct)
    dist_lu = dist(xcoord_lu, ycoord_lu)
    dist_ld = dist(xcoord_ld, ycoord_ld)
    dist_ru = dist(xcoord_ru, ycoord_ru)
    dist_rd = dist(xcoord_rd, ycoord_rd)
    outside = (dist_lu > r) & (dist_ld > r) & (dist_ru > r) & (dist_rd > r)
    inside = (dist_lu < r) & (dist_ld < r) & (dist_ru < r) & (dist_rd < r)
    a = T(0.5) + T(0.5) * (dist_ct - r) / prm.sqrt2dx
    a = var(a, T(0), T(1))
    return np.where(outside, T(1.0), np.where(inside, T(0.0), a)).astype(T)


# 2dvof.py:137-159
def set_init_F(s, ic):
    prm = s.prm
    T = prm.T
    nx, ny = prm.nx, prm.ny
    xi = prm.x[: nx + 2][:, None]
    yj = prm.y[: ny + 2][None, :]
    if ic == 1:
        x1, x2 = T(0.0), T(prm.Lx / 3)
        y1, y2 = T(0.0), T(prm.Ly / 2)
        m = (xi >= x1) & (xi <= x2) & (yj >= y1) & (yj <= y2)
        s.F[m] = T(1.0)
    elif ic == 2:
        r = T(prm.Lx / 12)
        cx, cy = T(prm.Lx / 2), T(2 * (prm.Lx / 12))
        s.F[...] = find_area(prm, cx, cy, r)
    elif ic == 3:
        r = T(prm.Lx / 12)
        cx, cy = T(prm.Lx / 2), T(prm.Ly - 3 * (prm.Lx / 12))
        s.F[...] = T(1.0) - find_area(prm, cx, cy, r)
        pool = np.broadcast_to(yj < T(prm.Ly * 0.37), s.F.shape)
        s.F[pool] = T(1.0)
    else:
        raise ValueError("ic must be 1, 2 or 3")


# 2dvof.py:162-189
def set_BC(s):
    nx, ny = s.prm.nx, s.prm.ny
    T = s.prm.T
    u, v, F, p, rho = s.u, s.v, s.F, s.p, s.rho
    # loop 1 over i in [0, nx+1]
    u[:, 0] = u[:, 1]
    v[:, 1] = T(0)
    F[:, 0] = F[:, 1]
    p[:, 0] = p[:, 1]
    rho[:, 0] = rho[:, 1]
    u[:, ny + 1] = u[:, ny]
    v[:, ny + 1] = T(0)
    F[:, ny + 1] = F[:, ny]
    p[:, ny + 1] = p[:, ny]
    rho[:, ny + 1] = rho[:, ny]
    # loop 2 over j in [0, ny+1]
    u[1, :] = T(0)
    v[0, :] = v[1, :]
    F[0, :] = F[1, :]
    p[0, :] = p[1, :]
    rho[0, :] = rho[1, :]
    u[nx + 1, :] = T(0)
    v[nx + 1, :] = v[nx, :]
    F[nx + 1, :] = F[nx, :]
    p[nx + 1, :] = p[nx, :]
    rho[nx + 1, :] = rho[nx, :]


# 2dvof.py:198-203
def cal_nu_rho(s):
    prm = s.prm
    T = prm.T
    Fc = var(T(0.0), T(1.0), s.F)
    s.rho[...] = prm.rho_g * (T(1) - Fc) + prm.rho_l * Fc
    s.nu[...] = prm.nu_l * Fc + prm.nu_g * (T(1.0) - Fc)


# 2dvof.py:283-309
def get_normal_young(s):
    prm = s.prm
    T = prm.T
    nx, ny = prm.nx, prm.ny
    F = s.F
    C = (slice(1, nx + 1), slice(1, ny + 1))

    def f(di, dj):
        return F[1 + di: nx + 1 + di, 1 + dj: ny + 1 + dj]

    cxn, cyn = prm.nrm_x, prm.nrm_y
    mx1 = cxn * (f(1, 1) + f(1, 0) - f(0, 1) - f(0, 0))
    my1 = cyn * (f(1, 1) - f(1, 0) + f(0, 1) - f(0, 0))
    mx2 = cxn * (f(1, 0) + f(1, -1) - f(0, 0) - f(0, -1))
    my2 = cyn * (f(1, 0) - f(1, -1) + f(0, 0) - f(0, -1))
    mx3 = cxn * (f(0, 0) + f(0, -1) - f(-1, 0) - f(-1, -1))
    my3 = cyn * (f(0, 0) - f(0, -1) + f(-1, 0) - f(-1, -1))
    mx4 = cxn * (f(0, 1) + f(0, 0) - f(-1, 1) - f(-1, 0))
    my4 = cyn * (f(0, 1) - f(0, 0) + f(-1, 1) - f(-1, 0))
    mxsum = (mx1 + mx2 + mx3 + mx4) / T(4)
    mysum = (my1 + my2 + my3 + my4) / T(4)
    small = (np.abs(mxsum) < prm.tiny) & (np.abs(mysum) < prm.tiny)
    magnitude = np.sqrt(mxsum * mxsum + mysum * mysum)
    with np.errstate(divide="ignore", invalid="ignore"):
        s.mx[C] = np.where(small, mxsum, mxsum / magnitude)
        s.my[C] = np.where(small, mysum, mysum / magnitude)
    # loop 2
    mx, my = s.mx, s.my
    s.kappa[C] = -(prm.kap_x * (mx[2: nx + 2, 1: ny + 1] - mx[0: nx, 1: ny + 1]) +
                   prm.kap_y * (my[1: nx + 1, 2: ny + 2] - my[1: nx + 1, 0: ny]))


# 2dvof.py:206-233
def advect_upwind(s):
    prm = s.prm
    T = prm.T
    nx, ny = prm.nx, prm.ny
    u, v, F, nu, rho, kappa = s.u, s.v, s.F, s.nu, s.rho, s.kappa
    dt, dxi, dyi, dxi2, dyi2 = prm.dt, prm.dxi, prm.dyi, prm.dxi2, prm.dyi2
    two = T(2)

    # loop 1: i in [2, nx], j in [1, ny]
    def a(arr, di, dj):
        return arr[2 + di: nx + 1 + di, 1 + dj: ny + 1 + dj]

    v_here = T(0.25) * (a(v, -1, 0) + a(v, -1, 1) + a(v, 0, 0) + a(v, 0, 1))
    dudx = np.where(a(u, 0, 0) > 0, (a(u, 0, 0) - a(u, -1, 0)) * dxi,
                    (a(u, 1, 0) - a(u, 0, 0)) * dxi)
    dudy = np.where(v_here > 0, (a(u, 0, 0) - a(u, 0, -1)) * dyi,
                    (a(u, 0, 1) - a(u, 0, 0)) * dyi)
    kappa_ave = (a(kappa, 0, 0) + a(kappa, -1, 0)) / T(2.0)
    fx_kappa = -prm.sigma * (a(F, 0, 0) - a(F, -1, 0)) * kappa_ave / prm.dx
    u_star_new = (
        a(u, 0, 0) + dt *
        (a(nu, 0, 0) * (a(u, -1, 0) - two * a(u, 0, 0) + a(u, 1, 0)) * dxi2
         + a(nu, 0, 0) * (a(u, 0, -1) - two * a(u, 0, 0) + a(u, 0, 1)) * dyi2
         - a(u, 0, 0) * dudx - v_here * dudy
         + prm.gx + fx_kappa * two / (a(rho, 0, 0) + a(rho, -1, 0)))
    )

    # loop 2: i in [1, nx], j in [2, ny]   (reads only u, v, ... not u_star)
    def b(arr, di, dj):
        return arr[1 + di: nx + 1 + di, 2 + dj: ny + 1 + dj]

    u_here = T(0.25) * (b(u, 0, -1) + b(u, 0, 0) + b(u, 1, -1) + b(u, 1, 0))
    dvdx = np.where(u_here > 0, (b(v, 0, 0) - b(v, -1, 0)) * dxi,
                    (b(v, 1, 0) - b(v, 0, 0)) * dxi)
    dvdy = np.where(b(v, 0, 0) > 0, (b(v, 0, 0) - b(v, 0, -1)) * dyi,
                    (b(v, 0, 1) - b(v, 0, 0)) * dyi)
    kappa_ave = (b(kappa, 0, 0) + b(kappa, 0, -1)) / T(2.0)
    fy_kappa = -prm.sigma * (b(F, 0, 0) - b(F, 0, -1)) * kappa_ave / prm.dy
    v_star_new = (
        b(v, 0, 0) + dt *
        (b(nu, 0, 0) * (b(v, -1, 0) - two * b(v, 0, 0) + b(v, 1, 0)) * dxi2
         + b(nu, 0, 0) * (b(v, 0, -1) - two * b(v, 0, 0) + b(v, 0, 1)) * dyi2
         - u_here * dvdx - b(v, 0, 0) * dvdy
         + prm.gy + fy_kappa * two / (b(rho, 0, 0) + b(rho, 0, -1)))
    )
    s.u_star[2: nx + 1, 1: ny + 1] = u_star_new
    s.v_star[1: nx + 1, 2: ny + 1] = v_star_new


# 2dvof.py:236-266 (one call = rhs + one Jacobi sweep + copy back)
def solve_p_jacobi(s):
    prm = s.prm
    T = prm.T
    nx, ny = prm.nx, prm.ny
    C = (slice(1, nx + 1), slice(1, ny + 1))
    p, rho, us, vs = s.p, s.rho, s.u_star, s.v_star
    rhs = rho[C] / prm.dt * \
        ((us[2: nx + 2, 1: ny + 1] - us[C]) * prm.dxi +
         (vs[1: nx + 1, 2: ny + 2] - vs[C]) * prm.dyi)
    i = np.arange(1, nx + 1)[:, None]
    j = np.arange(1, ny + 1)[None, :]
    zero = T(0.0)
    ae = np.where(i != prm.imax, prm.dxi2, zero).astype(T)
    aw = np.where(i != prm.imin, prm.dxi2, zero).astype(T)
    an = np.where(j != prm.jmax, prm.dyi2, zero).astype(T)
    a_s = np.where(j != prm.jmin, prm.dyi2, zero).astype(T)
    ap = T(-1.0) * (ae + aw + an + a_s)
    s.pt[C] = (rhs - ae * p[2: nx + 2, 1: ny + 1] - aw * p[0: nx, 1: ny + 1]
               - an * p[1: nx + 1, 2: ny + 2] - a_s * p[1: nx + 1, 0: ny]) / ap
    s.p[C] = s.pt[C]


# 2dvof.py:269-280
def update_uv(s):
    prm = s.prm
    T = prm.T
    nx, ny = prm.nx, prm.ny
    rho, p = s.rho, s.p
    A = (slice(2, nx + 1), slice(1, ny + 1))
    r = (rho[A] + rho[1: nx, 1: ny + 1]) * T(0.5)
    s.u[A] = s.u_star[A] - prm.dt / r * (p[A] - p[1: nx, 1: ny + 1]) * prm.dxi
    s.courant_violations += int(np.count_nonzero(s.u[A] * prm.dt > prm.cfl_x))
    B = (slice(1, nx + 1), slice(2, ny + 1))
    r = (rho[B] + rho[1: nx + 1, 1: ny]) * T(0.5)
    s.v[B] = s.v_star[B] - prm.dt / r * (p[B] - p[1: nx + 1, 1: ny]) * prm.dyi
    s.courant_violations += int(np.count_nonzero(s.v[B] * prm.dt > prm.cfl_y))


def _limiter(prm, q, pq):
    T = prm.T
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(pq > 0, _vmin(T(1), q / pq), T(0.0)).astype(T)


# 2dvof.py:321-382
def fct_x_sweep(s):
    prm = s.prm
    T = prm.T
    nx, ny = prm.nx, prm.ny
    C = (slice(1, nx + 1), slice(1, ny + 1))
    F, u = s.F, s.u
    dt, dx, dy, dxdy = prm.dt, prm.dx, prm.dy, prm.dxdy
    zero = T(0)

    def c(arr, di, dj=0):
        return arr[1 + di: nx + 1 + di, 1 + dj: ny + 1 + dj]

    # stage A
    dv = dxdy - prm.dtdy * (c(u, 1) - c(u, 0))
    fl_L = np.where(c(u, 0) >= 0, c(u, 0) * dt * c(F, -1), c(u, 0) * dt * c(F, 0))
    fr_L = np.where(c(u, 1) >= 0, c(u, 1) * dt * c(F, 0), c(u, 1) * dt * c(F, 1))
    ft_L = zero
    fb_L = zero
    Ftd = (c(F, 0) + (fl_L - fr_L + fb_L - ft_L) * dy / dxdy) * dx * dy / dv
    Ftd = np.where((Ftd > T(1.)) | (Ftd < 0), var(T(0), T(1), Ftd), Ftd)
    s.Ftd[C] = Ftd
    # stage B
    Ftd = s.Ftd
    fmax = _vmax(_vmax(c(Ftd, 0), c(Ftd, -1)), c(Ftd, 1))
    fmin = _vmin(_vmin(c(Ftd, 0), c(Ftd, -1)), c(Ftd, 1))
    fl_H = np.where(c(u, 0) <= 0, c(u, 0) * dt * c(F, -1), c(u, 0) * dt * c(F, 0))
    fr_H = np.where(c(u, 1) <= 0, c(u, 1) * dt * c(F, 0), c(u, 1) * dt * c(F, 1))
    s.ax[2: nx + 2, 1: ny + 1] = fr_H - fr_L
    s.ax[1: nx + 1, 1: ny + 1] = fl_H - fl_L
    s.ay[1: nx + 1, 2: ny + 2] = zero
    s.ay[1: nx + 1, 1: ny + 1] = zero
    ax, ay = s.ax, s.ay
    pp = _vmax(zero, c(ax, 0)) - _vmin(zero, c(ax, 1)) + _vmax(zero, c(ay, 0)) - _vmin(zero, c(ay, 0, 1))
    qp = (fmax - c(Ftd, 0)) * dx
    s.rp[C] = _limiter(prm, qp, pp)
    pm = _vmax(zero, c(ax, 1)) - _vmin(zero, c(ax, 0)) + _vmax(zero, c(ay, 0, 1)) - _vmin(zero, c(ay, 0))
    qm = (c(Ftd, 0) - fmin) * dx
    s.rm[C] = _limiter(prm, qm, pm)
    # stage C
    rp, rm = s.rp, s.rm
    s.cx[2: nx + 2, 1: ny + 1] = np.where(c(ax, 1) >= 0, _vmin(c(rp, 1), c(rm, 0)),
                                          _vmin(c(rp, 0), c(rm, 1)))
    s.cy[1: nx + 1, 2: ny + 2] = np.where(c(ay, 0, 1) >= 0, _vmin(c(rp, 0, 1), c(rm, 0)),
                                          _vmin(c(rp, 0), c(rm, 0, 1)))
    # stage D
    cx, cy = s.cx, s.cy
    dv = dxdy - prm.dtdy * (c(u, 1) - c(u, 0))
    Fn = c(Ftd, 0) - ((c(ax, 1) * c(cx, 1) -
                       c(ax, 0) * c(cx, 0) +
                       c(ay, 0, 1) * c(cy, 0, 1) -
                       c(ay, 0) * c(cy, 0)) / dy) * dx * dy / dv
    s.F[C] = var(T(0), T(1), Fn)


# 2dvof.py:385-448
def fct_y_sweep(s):
    prm = s.prm
    T = prm.T
    nx, ny = prm.nx, prm.ny
    C = (slice(1, nx + 1), slice(1, ny + 1))
    F, v = s.F, s.v
    dt, dx, dy, dxdy = prm.dt, prm.dx, prm.dy, prm.dxdy
    zero = T(0)

    def c(arr, di, dj=0):
        return arr[1 + di: nx + 1 + di, 1 + dj: ny + 1 + dj]

    # stage A
    dv = dxdy - prm.dtdx * (c(v, 0, 1) - c(v, 0))
    fl_L = zero
    fr_L = zero
    ft_L = np.where(c(v, 0, 1) >= 0, c(v, 0, 1) * dt * c(F, 0), c(v, 0, 1) * dt * c(F, 0, 1))
    fb_L = np.where(c(v, 0) >= 0, c(v, 0) * dt * c(F, 0, -1), c(v, 0) * dt * c(F, 0))
    Ftd = (c(F, 0) + (fl_L - fr_L + fb_L - ft_L) * dy / dxdy) * dx * dy / dv
    Ftd = np.where((Ftd > T(1.)) | (Ftd < 0), var(T(0), T(1), Ftd), Ftd)
    s.Ftd[C] = Ftd
    # stage B
    Ftd = s.Ftd
    fmax = _vmax(_vmax(c(Ftd, 0), c(Ftd, 0, -1)), c(Ftd, 0, 1))
    fmin = _vmin(_vmin(c(Ftd, 0), c(Ftd, 0, -1)), c(Ftd, 0, 1))
    ft_H = np.where(c(v, 0, 1) <= 0, c(v, 0, 1) * dt * c(F, 0), c(v, 0, 1) * dt * c(F, 0, 1))
    fb_H = np.where(c(v, 0) <= 0, c(v, 0) * dt * c(F, 0, -1), c(v, 0) * dt * c(F, 0))
    s.ax[2: nx + 2, 1: ny + 1] = zero
    s.ax[1: nx + 1, 1: ny + 1] = zero
    s.ay[1: nx + 1, 2: ny + 2] = ft_H - ft_L
    s.ay[1: nx + 1, 1: ny + 1] = fb_H - fb_L
    ax, ay = s.ax, s.ay
    pp = _vmax(zero, c(ax, 0)) - _vmin(zero, c(ax, 1)) + _vmax(zero, c(ay, 0)) - _vmin(zero, c(ay, 0, 1))
    qp = (fmax - c(Ftd, 0)) * dx
    s.rp[C] = _limiter(prm, qp, pp)
    pm = _vmax(zero, c(ax, 1)) - _vmin(zero, c(ax, 0)) + _vmax(zero, c(ay, 0, 1)) - _vmin(zero, c(ay, 0))
    qm = (c(Ftd, 0) - fmin) * dx
    s.rm[C] = _limiter(prm, qm, pm)
    # stage C
    rp, rm = s.rp, s.rm
    s.cx[2: nx + 2, 1: ny + 1] = np.where(c(ax, 1) >= 0, _vmin(c(rp, 1), c(rm, 0)),
                                          _vmin(c(rp, 0), c(rm, 1)))
    s.cy[1: nx + 1, 2: ny + 2] = np.where(c(ay, 0, 1) >= 0, _vmin(c(rp, 0, 1), c(rm, 0)),
                                          _vmin(c(rp, 0), c(rm, 0, 1)))
    # stage D
    cx, cy = s.cx, s.cy
    dv = dxdy - prm.dtdx * (c(v, 0, 1) - c(v, 0))
    Fn = c(Ftd, 0) - ((c(ax, 1) * c(cx, 1) -
                       c(ax, 0) * c(cx, 0) +
                       c(ay, 0, 1) * c(cy, 0, 1) -
                       c(ay, 0) * c(cy, 0)) / dy) * dx * dy / dv
    s.F[C] = var(T(0), T(1), Fn)


# 2dvof.py:312-318
def solve_VOF_rudman(s):
    if s.istep % 2 == 0:
        fct_y_sweep(s)
        fct_x_sweep(s)
    else:
        fct_x_sweep(s)
        fct_y_sweep(s)


# 2dvof.py:452-455
def post_process_f(s):
    T = s.prm.T
    s.F[...] = var(s.F, T(0), T(1))


# 2dvof.py:458-486: rgb_buf[I] = field[I // r], r = resolution[0] // nx = 2
def vis_field(s, which):
    prm = s.prm
    T = prm.T
    nx, ny = prm.nx, prm.ny
    a = (np.arange(2 * nx) // 2)[:, None]
    b = (np.arange(2 * ny) // 2)[None, :]
    if which == "vof":
        return s.F[a, b].copy()
    if which == "u":
        return s.u[a, b] / T(prm.Lx / 0.2)
    if which == "v":
        return s.v[a, b] / T(prm.Ly / 0.2)
    if which == "vnorm":
        return np.sqrt(s.u[a, b] * s.u[a, b] + s.v[a, b] * s.v[a, b]) / T(prm.Ly / 0.2)
    raise ValueError(which)


# 2dvof.py:488-492.  The reference's loop reaches u[imax+2, j], one row past the field (undefined
# in Taichi's release mode); that entry reads as 0 here.
def interp_velocity(s):
    prm = s.prm
    T = prm.T
    nx, ny = prm.nx, prm.ny
    V = np.zeros((nx + 2, ny + 2, 2), dtype=T)
    upad = np.vstack((s.u, np.zeros((1, ny + 2), dtype=T)))
    V[1: nx + 2, 1: ny + 1, 0] = (upad[1: nx + 2, 1: ny + 1] + upad[2: nx + 3, 1: ny + 1]) / T(2)
    V[1: nx + 2, 1: ny + 1, 1] = (s.v[1: nx + 2, 1: ny + 1] + s.v[1: nx + 2, 2: ny + 2]) / T(2)
    return V


# 2dvof.py:505-528 (solver part of the main loop)
def step(s, nsteps=1, jacobi_iters=10):
    for _ in range(nsteps):
        s.istep += 1
        cal_nu_rho(s)
        get_normal_young(s)
        advect_upwind(s)
        set_BC(s)
        for _ in range(jacobi_iters):
            solve_p_jacobi(s)
        update_uv(s)
        set_BC(s)
        solve_VOF_rudman(s)
        post_process_f(s)
        set_BC(s)


def new_state(nx, ny, ic=1, dtype=np.float64, coord_cast="f32", **kw):
    prm = Params(nx, ny, dtype=dtype, coord_cast=coord_cast, **kw)
    s = State(prm)
    set_init_F(s, ic)
    return s
