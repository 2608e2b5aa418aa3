// kernels/momentum.h -- k_momentum: normals + curvature + predictor + rhs in one pass
//
// Part of the gfx950 kernel set of the 2-D VOF hot path (see vof2d_kernels.h for the conventions:
// reference line citations, expression order, one wave = 64*V columns marching along i).
#pragma once
#include "common.h"
#include "plan.h"

namespace vof {

// ------------------------------------------------------------------ fused momentum + rhs
// get_normal_young (2dvof.py:283-309) + advect_upwind (:206-233) + the rhs of solve_p_jacobi
// (:239-241) in one pass: F, u, v -> u*, v*, rhs (6 array passes instead of 16).  mx, my and kappa
// live only in registers.  Pipeline along i with the newest F row r:
//   N: normals of row r-1   K: kappa of row r-2   P: u*, v* of row r-2   R: rhs of row r-3
// j+-1 neighbours of computed quantities (my, kappa, v*) come from adjacent lanes by shuffles,
// which invalidates 2 columns on each tile side (tiles overlap by 2*H, H = 2 rounded up to V).
// Never-written entries read as 0 exactly like the zero-initialised reference fields (S5):
// mx/my/kappa outside the interior, u* on wall faces, v* at j = 1 and j = ny+1.
// first-order upwind difference (:210-211, :223-224): pos ? (c - m) : (p - c).  Selecting the
// operands instead of the results performs the identical subtraction with half the arithmetic.
template <typename T>
__device__ __forceinline__ T upwind_diff(bool pos, T c, T m, T p) {
  const T a = pos ? c : p, b = pos ? m : c;
  return a - b;
}
// t / d for d > 0 (a sum of two densities).  Away from the interface the surface-tension force t
// is an exact zero and 0 / d = 0 with the sign of t, so the division is skipped (wave-level
// branch); otherwise it is the IEEE division.
template <typename T>
__device__ __forceinline__ T div_or_zero(T t, T d) {
  T r = t;
  if (t != (T)0) r = t / d;
  return r;
}

template <typename T>
__device__ __forceinline__ void normals_cell(const Consts<T>& c, T Fmm, T Fm0, T Fmp, T F0m, T F00, T F0p, T Fpm,
                                             T Fp0, T Fpp, T& ox, T& oy) {
  const T cxn = c.nrm_x, cyn = c.nrm_y;
  T mx1 = cxn * (Fpp + Fp0 - F0p - F00);
  T my1 = cyn * (Fpp - Fp0 + F0p - F00);
  T mx2 = cxn * (Fp0 + Fpm - F00 - F0m);
  T my2 = cyn * (Fp0 - Fpm + F00 - F0m);
  T mx3 = cxn * (F00 + F0m - Fm0 - Fmm);
  T my3 = cyn * (F00 - F0m + Fm0 - Fmm);
  T mx4 = cxn * (F0p + F00 - Fmp - Fm0);
  T my4 = cyn * (F0p - F00 + Fmp - Fm0);
  T mxsum = (mx1 + mx2 + mx3 + mx4) / (T)4;
  T mysum = (my1 + my2 + my3 + my4) / (T)4;
  if (dabs<T>(mxsum) < c.tiny && dabs<T>(mysum) < c.tiny) {
    ox = mxsum;
    oy = mysum;
  } else {
    T magnitude = dsqrt<T>(mxsum * mxsum + mysum * mysum);
    ox = mxsum / magnitude;
    oy = mysum / magnitude;
  }
}

// all V+2 values of a lane's row window are equal
template <typename T, int V>
__device__ __forceinline__ bool row_flat(const Row<T, V>& w) {
  bool f = w.l == w.c[0] && w.c[V - 1] == w.r;
#pragma unroll
  for (int q = 1; q < V; ++q) f = f && w.c[q] == w.c[0];
  return f;
}

// set_BC's ghost columns (2dvof.py:164-174: [i,0] = [i,1], [i,ny+1] = [i,ny]) applied to a loaded
// row window instead of to memory: a lane holds j0-1 | j0..j0+V-1 | j0+V, so the source column is
// always in the same lane.
template <typename T, int V>
__device__ __forceinline__ void mirror_ghost_cols(Row<T, V>& w, int j0, int ny) {
  if (j0 - 1 == 0) w.l = w.c[0];
#pragma unroll
  for (int q = 0; q < V; ++q)
    if (j0 + q == 0) w.c[q] = q == V - 1 ? w.r : w.c[q + 1];
#pragma unroll
  for (int q = 0; q < V; ++q)
    if (j0 + q == ny + 1) w.c[q] = q == 0 ? w.l : w.c[q - 1];
  if (j0 + V == ny + 1) w.r = w.c[V - 1];
}

// ------------------------------------------------------------------ one row of the pipeline, shared by k_momentum and k_tm
// The register window of the march and ONE iteration of its row loop: the newest rows come in (F row r, u / v rows r-1, ghost
// columns already mirrored), u*, v* of row r-2 and the rhs of row r-3 come out, the windows shift.  Where the rows come from
// (global loads, prefetched an iteration ahead: k_momentum; the LDS ring of a wave pair: k_tm's momentum wave) and where the
// results go (exec-masked or range-checked stores) is the caller's business -- one source for the arithmetic (round 6; rounds
// 4-5 kept a copy per kernel).
//   IN (an interior pair of k_tm): every row the iteration touches lies in [3, nx - 1] and every column of the tile in [2, ny] --
//   the row / column selects around the results are constants.  ABL: timing-only ablations of the diagnostic build.
template <typename T, int V>
struct MomentumWindow {
  Row<T, V> F2, F1;            // F rows r-2, r-1 (become r-3.. after the shift)
  T F3c[V];                    // F row r-3, centre columns
  Row<T, V> u3, u2, v3, v2;    // u, v rows r-3, r-2 (row r-1 arrives with the iteration)
  T mx2[V], mx3[V], my2[V];    // mx rows r-2, r-3; my row r-2
  T k3[V];                     // kappa row r-3
  T us3[V], vs3[V];            // u*, v* row r-3
  T rho3[V];                   // rho(F) row r-3 (rho is a pure function of F[i,j], :201-202)
  bool flat2, flat1;           // rows r-2, r-1: all V + 2 values a lane sees are equal
  // Rows of uniform F (gas, or liquid away from the interface: nine rows in ten of a dam-break) -- wave-level history of
  // `flat` (the three newest rows uniform and equal), bit k = the iteration k before this one.  What it lets a row skip is
  // exact: the skipped arithmetic would produce the same zeros / the same rho, nu from the same F.
  unsigned flat_hist;
  bool zero_force_ok;          // the force terms of a flat window are +-0; they enter u*, v* as (... + gx) + fx: the sum in front is never -0 unless gx is

  static __device__ __forceinline__ void zero_row(Row<T, V>& w) {
    w.l = w.r = (T)0;
#pragma unroll
    for (int q = 0; q < V; ++q) w.c[q] = (T)0;
  }
  // u, v rows below the chunk's first row - 1 are never used by a stored value (the first stored u*, v* row reads the rows
  // around it; F needs three rows more for the normals behind kappa): the window starts from zeros.  F2, F1 are the
  // caller's to fill (set_F) before the first iteration.
  __device__ __forceinline__ void init(const Consts<T>& c) {
    zero_row(u3); zero_row(u2); zero_row(v3); zero_row(v2);
    zero_row(F2); zero_row(F1);
#pragma unroll
    for (int q = 0; q < V; ++q) F3c[q] = mx2[q] = mx3[q] = my2[q] = k3[q] = us3[q] = vs3[q] = rho3[q] = (T)0;
    flat2 = flat1 = true;
    flat_hist = 0u;
    zero_force_ok = !(c.gx == (T)0 && __builtin_signbit(c.gx)) && !(c.gy == (T)0 && __builtin_signbit(c.gy));
  }
  __device__ __forceinline__ void set_F(const Row<T, V>& f2, const Row<T, V>& f1) {
    F2 = f2; F1 = f1;
    flat2 = row_flat<T, V>(F2);
    flat1 = row_flat<T, V>(F1);
  }

  // store_uv(us2, vs2) is called between the predictor and the rhs stage (where the kernels have always issued those stores);
  // the rhs row is the caller's to store after the call
  template <bool IN, int ABL = 0, typename StoreUV>
  __device__ __forceinline__ void step(const Consts<T>& c, int r, int ilo, int ihi, int j0, int ny, const bool (&dom)[V],
                                       const Row<T, V>& F0, const Row<T, V>& u1, const Row<T, V>& v1, T (&rhs3)[V], bool want_rhs,
                                       StoreUV&& store_uv) {
    const T dt = c.dt, dxi = c.dxi, dyi = c.dyi, dxi2 = c.dxi2, dyi2 = c.dyi2;
    // ---- N: normals of row r-1 (:285-306)
    const bool okN = IN || ((r - 1) >= ilo && (r - 1) <= ihi);
    T mx1[V], my1[V];
    // Away from the interface all 3 x (V+2) values of F a lane sees are equal; every corner
    // difference of :287-294 is then an exact zero and (mx, my) = (0, 0).  When that holds for the
    // whole wave the stage is skipped (flat2 / flat1 cache the per-row test, one row is new per step).
    const bool flat0 = row_flat<T, V>(F0);
    const bool flat = flat2 && flat1 && flat0 && F2.c[0] == F1.c[0] && F1.c[0] == F0.c[0];
    const bool wflat = __all(flat);
#ifdef VOF_NO_FLAT_SHORTCUTS      // (A/B builds: make variant NAME=noflat EXTRA=-DVOF_NO_FLAT_SHORTCUTS)
    flat_hist = 0u;
#else
    flat_hist = __builtin_amdgcn_readfirstlane((flat_hist << 1) | (wflat ? 1u : 0u));
#endif
    VOF_STAT(0);
    if (wflat) {
      VOF_STAT(1);
#pragma unroll
      for (int q = 0; q < V; ++q) mx1[q] = my1[q] = (T)0;
    } else {
#pragma unroll
      for (int q = 0; q < V; ++q) {
        T ox, oy;
        normals_cell<T>(c, left_of(F2, q), F2.c[q], right_of(F2, q), left_of(F1, q), F1.c[q], right_of(F1, q),
                        left_of(F0, q), F0.c[q], right_of(F0, q), ox, oy);
        mx1[q] = (okN && dom[q]) ? ox : (T)0;
        my1[q] = (okN && dom[q]) ? oy : (T)0;
      }
    }
    // ---- K: kappa of row r-2 (:307-309)
    const bool okK = IN || ((r - 2) >= ilo && (r - 2) <= ihi);
    T k2[V];
    if ((flat_hist & 7u) == 7u) {
      // flat for three iterations: mx1, mx3 and every my2 of the wave are the +0 the flat branch above assigned
      const T kk = -(c.kap_x * ((T)0 - (T)0) + c.kap_y * ((T)0 - (T)0));
#pragma unroll
      for (int q = 0; q < V; ++q) k2[q] = (okK && dom[q]) ? kk : (T)0;
    } else {
      const T myl = lane_up(my2[V - 1]), myr = lane_dn(my2[0]);
#pragma unroll
      for (int q = 0; q < V; ++q) {
        const T yr = q == V - 1 ? myr : my2[q + 1], yl = q == 0 ? myl : my2[q - 1];
        const T kk = -(c.kap_x * (mx1[q] - mx3[q]) + c.kap_y * (yr - yl));
        k2[q] = (okK && dom[q]) ? kk : (T)0;
      }
    }
    // ---- P: u*, v* of row i = r-2 (:206-233)
    const int i = r - 2;
    const bool okP = IN || (i >= ilo && i <= ihi);
    T us2[V], vs2[V];
    T rho2[V], nu2[V];
    T rho2l;
    if ((flat_hist & 3u) != 0u) {   // row r-2 is uniform (it belongs to a flat window): one rho, one nu
      const T rf = rho_of(c, F2.c[0]), nf = nu_of(c, F2.c[0]);
#pragma unroll
      for (int q = 0; q < V; ++q) { rho2[q] = rf; nu2[q] = nf; }
      rho2l = rf;
    } else {
#pragma unroll
      for (int q = 0; q < V; ++q) { rho2[q] = rho_of(c, F2.c[q]); nu2[q] = nu_of(c, F2.c[q]); }
      rho2l = rho_of(c, F2.l);
    }
    // Surface tension (:213-214, :225-226): force = (-sigma * dF * kappa_ave / dx) * 2 / (rho + rho').
    // Away from the interface dF or kappa_ave is an exact zero and so is the force; one wave-level
    // test covers the 2 V quotient pairs of the lane, and the exact divisions run only behind it.
    T fxf[V], fyf[V];
    bool any_force = false;
    if ((flat_hist & 2u) != 0u && zero_force_ok) {
      // rows r-3, r-2 (and r-1) uniform and equal: F00 - Fm0 and F00 - F0m are +0, the force numerators +-0 whatever kappa is
#pragma unroll
      for (int q = 0; q < V; ++q) fxf[q] = fyf[q] = (T)0;
    } else {
      const T kl = lane_up(k2[V - 1]);
#pragma unroll
      for (int q = 0; q < V; ++q) {
        const T F00 = F2.c[q], Fm0 = F3c[q], F0m = left_of(F2, q);
        const T k00 = k2[q], km0 = k3[q], k0m = q == 0 ? kl : k2[q - 1];
        fxf[q] = -c.sigma * (F00 - Fm0) * ((k00 + km0) / (T)2.0);
        fyf[q] = -c.sigma * (F00 - F0m) * ((k00 + k0m) / (T)2.0);
        any_force = any_force || fxf[q] != (T)0 || fyf[q] != (T)0;
      }
    }
    if (!__any(any_force)) VOF_STAT(2);
    if (any_force) {
#pragma unroll
      for (int q = 0; q < V; ++q) {
        const T rho00 = rho2[q], rhom0 = rho3[q], rho0m = q == 0 ? rho2l : rho2[q - 1];
        const T fnum[2] = {fxf[q], fyf[q]}, fden[2] = {c.dx, c.dy}, finv[2] = {c.inv_dx, c.inv_dy};
        T fk[2];
        div_by_const_v<T, 2, true>(fk, fnum, fden, finv);
        fxf[q] = div_or_zero<T>(fk[0] * (T)2, rho00 + rhom0);
        fyf[q] = div_or_zero<T>(fk[1] * (T)2, rho00 + rho0m);
      }
    }
#pragma unroll
    for (int q = 0; q < V; ++q) {
      const T u00 = u2.c[q], um0 = u3.c[q], up0 = u1.c[q], u0m = left_of(u2, q), u0p = right_of(u2, q);
      const T upm = left_of(u1, q);
      const T v00 = v2.c[q], vm0 = v3.c[q], vp0 = v1.c[q], v0m = left_of(v2, q), v0p = right_of(v2, q);
      const T vmp = right_of(v3, q);
      const T nu00 = nu2[q];
      T ou, ov;
      {
        T v_here = (T)0.25 * (vm0 + vmp + v00 + v0p);
        T dudx = upwind_diff<T>(u00 > 0, u00, um0, up0) * dxi;      // (u00-um0)*dxi or (up0-u00)*dxi
        T dudy = upwind_diff<T>(v_here > 0, u00, u0m, u0p) * dyi;
        ou = (u00 + dt * (nu00 * (um0 - (T)2 * u00 + up0) * dxi2 + nu00 * (u0m - (T)2 * u00 + u0p) * dyi2 -
                          u00 * dudx - v_here * dudy + c.gx + fxf[q]));
        if constexpr ((ABL & ABL_NO_UPRED) != 0) ou = u00 + dt * (um0 + c.gx + fxf[q]);   // (timing only)
      }
      {
        T u_here = (T)0.25 * (u0m + u00 + upm + up0);
        T dvdx = upwind_diff<T>(u_here > 0, v00, vm0, vp0) * dxi;
        T dvdy = upwind_diff<T>(v00 > 0, v00, v0m, v0p) * dyi;
        ov = (v00 + dt * (nu00 * (vm0 - (T)2 * v00 + vp0) * dxi2 + nu00 * (v0m - (T)2 * v00 + v0p) * dyi2 -
                          u_here * dvdx - v00 * dvdy + c.gy + fyf[q]));
        if constexpr ((ABL & ABL_NO_VPRED) != 0) ov = v00 + dt * (vm0 + c.gy + fyf[q]);   // (timing only)
      }
      const int j = j0 + q;
      us2[q] = (okP && (IN || i >= 2) && dom[q]) ? ou : (T)0;           // u* exists on i in [2, nx]
      vs2[q] = (okP && (IN || (j >= 2 && j <= ny))) ? ov : (T)0;        // v* exists on j in [2, ny]
    }
    store_uv(us2, vs2);
    // ---- R: rhs of row r-3 (:239-241)
    if (want_rhs) {
      const T vsr = lane_dn(vs3[0]);
#pragma unroll
      for (int q = 0; q < V; ++q) {
        const T vright = q == V - 1 ? vsr : vs3[q + 1];
        // rho lies in [rho_g, rho_l] (var clamps F, :192-196): always inside the fast window
        rhs3[q] = div_by_const_inrange<T>(rho3[q], c.dt, c.inv_dt) *
                  ((us2[q] - us3[q]) * c.dxi + (vright - vs3[q]) * c.dyi);
      }
    }
    // ---- shift the windows
#pragma unroll
    for (int q = 0; q < V; ++q) {
      F3c[q] = F2.c[q];
      mx3[q] = mx2[q]; mx2[q] = mx1[q]; my2[q] = my1[q];
      k3[q] = k2[q];
      us3[q] = us2[q]; vs3[q] = vs2[q];
      rho3[q] = rho2[q];
    }
    F2 = F1; F1 = F0;
    flat2 = flat1; flat1 = flat0;
    u3 = u2; u2 = u1;
    v3 = v2; v2 = v1;
  }
};

// BS (even ny, fields below 2 GiB: chosen by the launch wrapper): every memory instruction of the row loop is
// unconditional -- the loads run one clamped row past the chunk, the stores are range-checked buffer stores
// (store_buf_nt) whose lanes outside [jlo, jhi] and rows outside the chunk are dropped by the hardware.  The
// compiler then counts them, and the wait for the rows requested an iteration ago no longer includes the three
// stores issued since: 187 -> 172 us at 4096^2 fp64 (profiles/r04_ab_buffer_stores.log).
template <typename T, int V, bool BS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void k_momentum(Geom g, Consts<T> c, const T* __restrict__ F,
                                                   const T* __restrict__ u, const T* __restrict__ v,
                                                   T* __restrict__ us, T* __restrict__ vs, T* __restrict__ rhs,
                                                   int R, int ntt, int virt, TbPlan tp, int first, int last) {
  // rows [first, last] (within [g.ilo, g.ihi]) are produced; the domain of the zero-ghost conventions stays [g.ilo, g.ihi]
  // the launch's FIRST block is the planner of this step's k_jacobi_tb launches (see tb_make_plan): it
  // starts with the launch and runs beside the other blocks (as the last block it would start when the
  // last slots free up and add its few microseconds to the kernel's tail)
  const int plan_blocks = tp.masks != nullptr ? 1 : 0;
  if (plan_blocks && blockIdx.x == 0) {
    __shared__ TbPlanShared plan_sh;
    tb_make_plan(g, tp, plan_sh);
    return;
  }
  // virt (full-domain fused steps, DESIGN.md "virtual ghosts"): the previous step did not run
  // set_BC; the ghost cells this kernel reads -- F's ghost rows and columns, v's ghost rows, u's
  // ghost columns -- are formed from the interior cells set_BC would have copied (:164-189).
  constexpr int W = 64 * V;
  constexpr int H = TileHalo::momentum;
  static_assert(H >= 2 && H % V == 0, "two columns of each side are invalid after the cross-lane stages");
  WaveTimer wt_(WT_MOMENTUM);
  constexpr int STRIDE = W - 2 * H;
  const int wave = ((int)blockIdx.x - plan_blocks) * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int tj = wave % ntt, ch = wave / ntt;
  const int c0 = 1 - H + tj * STRIDE;
  const int j0 = c0 + lane * V;
  const int ra = first + ch * R;
  if (ra > last) return;  // wave-uniform
  const int rb = ra + R - 1 < last ? ra + R - 1 : last;
  const int ny = g.ny, ilo = g.ilo, ihi = g.ihi;
  const int jlo = c0 + H > 1 ? c0 + H : 1;
  const int jhi = c0 + W - H - 1 < ny ? c0 + W - H - 1 : ny;
  bool dom[V];
#pragma unroll
  for (int q = 0; q < V; ++q) dom[q] = (j0 + q) >= 1 && (j0 + q) <= ny;
  const bool vlo = virt && g.wall_lo, vhi = virt && g.wall_hi;
  auto rowptr = [&](const T* base, int r) {
    const int rc = r < g.row_lo ? g.row_lo : (r > g.row_hi ? g.row_hi : r);
    return base + (size_t)(rc - g.row_lo) * (size_t)g.pitch + (size_t)(g.col0 + j0);
  };
  // ghost rows 0 / nx+1 of F and v mirror rows 1 / nx (:176-189); u's are stored (u[nx+1] = 0)
  auto mirrow = [&](int r) { return (vlo && r == 0) ? 1 : ((vhi && r == g.nx + 1) ? g.nx : r); };
  const bool edge_cols = virt && (c0 - 1 <= 0 || c0 + W >= ny + 1);   // wave-uniform: the tile holds a ghost column
  // (the ghost columns are mirrored when a row is taken into use, not when it is loaded: the
  // prefetched rows stay in flight for a whole iteration)
  auto load_F = [&](Row<T, V>& w, int r) { load_row<T, V>(w, rowptr(F, mirrow(r))); };
  auto load_u = [&](Row<T, V>& w, int r) { load_row<T, V>(w, rowptr(u, r)); };
  auto load_v = [&](Row<T, V>& w, int r) { load_row<T, V>(w, rowptr(v, mirrow(r))); };
  // the register window and the arithmetic of an iteration: MomentumWindow (shared with k_tm)
  MomentumWindow<T, V> win;
  win.init(c);
  const int r0 = ra - 1, r1 = rb + 3;
  const T* const us_tile = us + (int64_t)(g.col0 + c0);    // BS: (wave-uniform) first column of the tile in stored row row_lo
  const T* const vs_tile = vs + (int64_t)(g.col0 + c0);
  const T* const rhs_tile = rhs + (int64_t)(g.col0 + c0);
  const int voff_st = (j0 >= jlo && j0 + V - 1 <= jhi) ? lane * (int)(V * sizeof(T)) : kBufSkip;   // (even ny: no lane is cut by jlo / jhi)
  {
    Row<T, V> f2, f1;
    load_F(f2, r0 - 2);
    load_F(f1, r0 - 1);
    if (edge_cols) {
      mirror_ghost_cols<T, V>(f2, j0, ny);
      mirror_ghost_cols<T, V>(f1, j0, ny);
    }
    win.set_F(f2, f1);
  }
  Row<T, V> Fn, un, vn;  // prefetched: F row r, u / v row r-1
  load_F(Fn, r0);
  MomentumWindow<T, V>::zero_row(un); MomentumWindow<T, V>::zero_row(vn);   // (row ra-2: unused, see MomentumWindow::init)
  for (int r = r0; r <= r1; ++r) {
    Row<T, V> F0 = Fn, u1 = un;
    const Row<T, V> v1 = vn;
    if (BS || r < r1) {    // (BS: past the chunk's last row a clamped row is loaded and never used)
      load_F(Fn, r + 1);
      load_u(un, r);
      load_v(vn, r);
    }
    if (edge_cols) {   // (after the prefetch has been issued)
      mirror_ghost_cols<T, V>(F0, j0, ny);
      mirror_ghost_cols<T, V>(u1, j0, ny);
    }
    const int i = r - 2, i3 = r - 3;
    T out[V];
    win.template step<false>(c, r, ilo, ihi, j0, ny, dom, F0, u1, v1, out, BS || (i3 >= ra && i3 <= rb), [&](const T (&us2)[V], const T (&vs2)[V]) {
      if constexpr (BS) {
        const bool rowok = i >= ra && i <= rb;
        const int vo = rowok ? voff_st : kBufSkip;
        const int so = rowok ? (int)((int64_t)(i - g.row_lo) * g.pitch * (int64_t)sizeof(T)) : 0;   // (a dropped row keeps an in-field offset)
        store_buf_nt<T, V>(us_tile, vo, so, us2);    // (row 1 and column 1 of v* carry the zeros the never-written entries hold)
        store_buf_nt<T, V>(vs_tile, vo, so, vs2);
      } else if (i >= ra && i <= rb) {
        if (i >= 2) store_s<T, V>(us + at(g, i, j0), us2, j0, jlo, jhi);
        store_s<T, V>(vs + at(g, i, j0), vs2, j0, jlo > 2 ? jlo : 2, jhi);
      }
    });
    if constexpr (BS)
      store_buf_nt<T, V>(rhs_tile, (i3 >= ra && i3 <= rb) ? voff_st : kBufSkip,
                         (i3 >= ra && i3 <= rb) ? (int)((int64_t)(i3 - g.row_lo) * g.pitch * (int64_t)sizeof(T)) : 0, out);
    else if (i3 >= ra && i3 <= rb)
      store_s<T, V>(rhs + at(g, i3, j0), out, j0, jlo, jhi);
  }
}

}  // namespace vof
