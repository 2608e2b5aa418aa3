// kernels/transport.h -- FCT sweeps (k_fct_x, k_fct_y) and the fused transport k_transport
//
// Part of the gfx950 kernel set of the 2-D VOF hot path (see vof2d_kernels.h for the conventions:
// reference line citations, expression order, one wave = 64*V columns marching along i).
#pragma once
#include "common.h"

namespace vof {

// ------------------------------------------------------------------ FCT
// Shared per-face / per-cell arithmetic of fct_x_sweep / fct_y_sweep.
// Face f between cells f-1 and f carries velocity w:  (2dvof.py:325-326, 342-343; S4)
//   L(f) = (w*dt) * (w >= 0 ? F[f-1] : F[f])     low-order (donor) flux
//   H(f) = (w*dt) * (w <= 0 ? F[f-1] : F[f])     high-order (downwind) flux
//   a(f) = H(f) - L(f)                           anti-diffusive flux (ax / ay)
template <typename T>
__device__ __forceinline__ void fct_face(T w, T dt, T Fm, T Fp, T& L, T& a) {
  const T wd = w * dt;
  L = w >= 0 ? wd * Fm : wd * Fp;
  const T H = w <= 0 ? wd * Fm : wd * Fp;
  a = H - L;
}
// stage A (:324-331 / :388-395): Ftd from F, the low-order fluxes through the
// cell's lower (Llo) and upper (Lhi) face, and dv.
//   x-sweep: flux = fl_L - fr_L + 0 - 0 ; y-sweep: flux = 0 - 0 + fb_L - ft_L  (same value: Llo - Lhi)
template <typename T>
__device__ __forceinline__ T fct_ftd(const Consts<T>& c, T F, T Llo, T Lhi, T dv) {
  T ftd = (F + div_by_const<T, true>((Llo - Lhi) * c.dy, c.dxdy, c.inv_dxdy)) * c.dx * c.dy / dv;
  if (ftd > (T)1. || ftd < 0) ftd = var3((T)0, (T)1, ftd);
  return ftd;
}
// stage B limiter ratios (:351-363 / :417-429); alo / ahi = anti-diffusive flux
// through the cell's lower / upper face; the other direction's terms are exact zeros.
template <typename T>
__device__ __forceinline__ void fct_ratios(const Consts<T>& c, T ftd, T ftd_m, T ftd_p, T alo, T ahi, T& rp, T& rm) {
  const T Z = (T)0;
  T fmax = vmax(vmax(ftd, ftd_m), ftd_p);
  T fmin = vmin(vmin(ftd, ftd_m), ftd_p);
  T pp = vmax(Z, alo) - vmin(Z, ahi);
  T qp = (fmax - ftd) * c.dx;  // dx in both sweeps (:417)
  rp = pp > 0 ? vmin((T)1, qp / pp) : (T)0.0;
  T pm = vmax(Z, ahi) - vmin(Z, alo);
  T qm = (ftd - fmin) * c.dx;
  rm = pm > 0 ? vmin((T)1, qm / pm) : (T)0.0;
}
// stage C (:365-374 / :431-440): limiter of face f between cells f-1 (m) and f (p)
template <typename T>
__device__ __forceinline__ T fct_climit(T a, T rp_m, T rm_m, T rp_p, T rm_p) {
  return a >= 0 ? vmin(rp_p, rm_m) : vmin(rp_m, rm_p);
}
// stage D (:376-382 / :442-448) + optional fused post_process_f (:452-455)
template <typename T, bool POST>
__device__ __forceinline__ T fct_final(const Consts<T>& c, T ftd, T alo, T clo, T ahi, T chi, T dv) {
  T f = ftd - div_by_const<T, true>(ahi * chi - alo * clo, c.dy, c.inv_dy) * c.dx * c.dy / dv;
  f = var3((T)0, (T)1, f);
  if (POST) f = var3(f, (T)0, (T)1);
  return f;
}

// stage D where both anti-diffusive fluxes of the cell are exact zeros: F = var(0, 1, F~ - 0)
template <typename T, bool POST>
__device__ __forceinline__ T fct_clamp(T ftd) {
  T f = var3((T)0, (T)1, ftd);
  if (POST) f = var3(f, (T)0, (T)1);
  return f;
}

// 2dvof.py:321-382 fct_x_sweep, the four barrier-separated loops fused into
// one pass: each lane marches along i (the sweep direction) with a 3-row-deep
// software pipeline (face -> Ftd -> rp/rm -> cx -> F').  Out of place: reads
// F, writes Fn (the twin); the host swaps the two pointers afterwards.
// Zero-ghost semantics (S5): Ftd, rp, rm outside [ilo, ihi] and cx at face
// ilo read as 0, exactly what the never-written ghost entries hold.
// update_uv (2dvof.py:269-280) for one cell, shared by k_correct's fused forms below: the same
// expressions in the same order.  rho_c / rho_m: density of the cell and of its lower neighbour in
// the component's direction; pc / pm likewise for p.
template <typename T>
__device__ __forceinline__ T corrected_velocity(const Consts<T>& c, T star, T rho_c, T rho_m, T pc, T pm, T di) {
  const T r = (rho_c + rho_m) * (T)0.5;
  return star - c.dt / r * (pc - pm) * di;
}

// fct_y_sweep for one row segment: a wave's 64*V consecutive cells, valid for the inner columns
// [c0+4, c0+W-5] (the j+-3 dependency is resolved across lanes; tiles overlap by 8 columns)
// IN: every column the wave holds lies in [2, ny] (an interior tile of a pair kernel): the column tests are constants
template <typename T, int V, bool POST, bool IN = false>
__device__ __forceinline__ void fct_y_row(const Consts<T>& c, int j0, int ny, const T (&Fz)[V], const T (&vz)[V],
                                          T (&out)[V]) {
  const T Fl = lane_up(Fz[V - 1]);
  T L[V], a[V];
#pragma unroll
  for (int q = 0; q < V; ++q) fct_face<T>(vz[q], c.dt, q == 0 ? Fl : Fz[q - 1], Fz[q], L[q], a[q]);
  const T Ln = lane_dn(L[0]), an_ = lane_dn(a[0]), vn = lane_dn(vz[0]);
  T td[V], dv[V];
#pragma unroll
  for (int q = 0; q < V; ++q) {
    const int j = j0 + q;
    dv[q] = c.dxdy - c.dtdx * ((q == V - 1 ? vn : vz[q + 1]) - vz[q]);
    td[q] = (IN || (j >= 1 && j <= ny)) ? fct_ftd<T>(c, Fz[q], L[q], q == V - 1 ? Ln : L[q + 1], dv[q]) : (T)0;
  }
  // Stages B, C, D.  Where every anti-diffusive flux the wave holds is an exact zero (F uniform along the sweep:
  // the bulk of either phase) the limiter ratios of :417-429 are 0 (pp = pm = 0), so are the face limiters, and stage
  // D subtracts (0 / dy) * dx * dy / dv = 0: the new F is the clamped F~.  One wave-level test replaces two IEEE
  // divisions, the limiter and the division by dv per cell.
  bool anz = an_ != (T)0;
#pragma unroll
  for (int q = 0; q < V; ++q) anz = anz || a[q] != (T)0;
  VOF_STAT(10);
  if (!__any(anz)) {
    VOF_STAT(11);
#pragma unroll
    for (int q = 0; q < V; ++q) out[q] = fct_clamp<T, POST>(td[q]);
    return;
  }
  const T tl = lane_up(td[V - 1]), tr = lane_dn(td[0]);
  T rp[V], rm[V];
#pragma unroll
  for (int q = 0; q < V; ++q) {
    const int j = j0 + q;
    rp[q] = rm[q] = (T)0;
    if (IN || (j >= 1 && j <= ny))
      fct_ratios<T>(c, td[q], q == 0 ? tl : td[q - 1], q == V - 1 ? tr : td[q + 1], a[q],
                    q == V - 1 ? an_ : a[q + 1], rp[q], rm[q]);
  }
  const T rpl = lane_up(rp[V - 1]), rml = lane_up(rm[V - 1]);
  T cy[V];
#pragma unroll
  for (int q = 0; q < V; ++q) {
    const int j = j0 + q;  // face j between cells j-1 and j; written for j in [2, ny+1]
    cy[q] = (IN || (j >= 2 && j <= ny + 1))
                ? fct_climit<T>(a[q], q == 0 ? rpl : rp[q - 1], q == 0 ? rml : rm[q - 1], rp[q], rm[q])
                : (T)0;
  }
  const T cn = lane_dn(cy[0]);
#pragma unroll
  for (int q = 0; q < V; ++q)
    out[q] = fct_final<T, POST>(c, td[q], a[q], cy[q], q == V - 1 ? an_ : a[q + 1], q == V - 1 ? cn : cy[q + 1],
                                dv[q]);
}

// fct_x_sweep as a pipeline along i (see k_fct_x): push row r of F and of the face velocity u,
// receive row r-3 of the swept F.  State indices are relative to the newest row.
template <typename T, int V>
struct FctXPipe {
  T F1[V], u1[V], L1[V], a1[V], a2[V], a3[V], t2[V], t3[V], d2[V], d3[V], rp3[V], rm3[V], c3[V];
  int zrows;
  // wave-uniform flags: some lane of the wave holds a non-zero anti-diffusive flux on the face row (nz1 / nz2 / nz3
  // for a1 / a2 / a3), a non-zero limiter ratio in the cell row behind (nzr3 for rp3, rm3).  Where a flag is false
  // the stage that would multiply or divide by those zeros is skipped: its results are the same exact zeros.
  bool nz1, nz2, nz3, nzr3;
  __device__ __forceinline__ void init(const T (&Fm)[V]) {  // Fm = F[row before the first pushed row]
#pragma unroll
    for (int q = 0; q < V; ++q) {
      F1[q] = Fm[q];
      u1[q] = L1[q] = a1[q] = a2[q] = a3[q] = t2[q] = t3[q] = rp3[q] = rm3[q] = c3[q] = (T)0;
      d2[q] = d3[q] = (T)1;
    }
    zrows = 0;
    nz1 = nz2 = nz3 = nzr3 = false;
  }
  // zero_row: Fr is an exact zero on every lane of the wave (the caller has looked at the row anyway)
  // IN: rows r - 3 .. r all lie strictly inside [ilo, ihi] (an interior chunk of a pair kernel): the row tests are constants
  template <bool POST, bool IN = false>
  __device__ __forceinline__ void push(const Consts<T>& c, int r, int ilo, int ihi, const T (&Fr)[V],
                                       const T (&ur)[V], T (&out)[V], bool zero_row) {
    zrows = zero_row ? zrows + 1 : 0;
    if (zrows >= 7) VOF_STAT(7);
    if (zrows >= 7) {  // the whole dependency window F[r-6..r] of the wave is zero: every output is
#pragma unroll
      for (int q = 0; q < V; ++q) {
        const T dv1 = c.dxdy - c.dtdy * (ur[q] - u1[q]);
        out[q] = (T)0;
        F1[q] = Fr[q]; u1[q] = ur[q]; L1[q] = (T)0;
        a3[q] = a2[q] = a1[q] = (T)0;
        t3[q] = t2[q] = (T)0;
        d3[q] = d2[q]; d2[q] = dv1;
        rp3[q] = rm3[q] = c3[q] = (T)0;
      }
      nz1 = nz2 = nz3 = nzr3 = false;
      return;
    }
    const int i1 = r - 1, i2 = r - 2;
    const bool in1 = IN || (i1 >= ilo && i1 <= ihi), in2 = IN || (i2 >= ilo && i2 <= ihi), inc2 = IN || (i2 > ilo && i2 <= ihi + 1);   // wave-uniform
    T Lr[V], ar[V], dv1[V], tn[V];
    bool nzr = false;
#pragma unroll
    for (int q = 0; q < V; ++q) {
      fct_face<T>(ur[q], c.dt, F1[q], Fr[q], Lr[q], ar[q]);
      nzr = nzr || ar[q] != (T)0;
      dv1[q] = c.dxdy - c.dtdy * (ur[q] - u1[q]);
      tn[q] = (T)0;
    }
    const bool nz0 = __any(nzr);
    if (in1) {
#pragma unroll
      for (int q = 0; q < V; ++q) tn[q] = fct_ftd<T>(c, F1[q], L1[q], Lr[q], dv1[q]);
    }
    // stage B of row r-2 (faces a2 below, a1 above): all fluxes zero -> pp = pm = 0 -> both ratios 0
    T rp2[V], rm2[V];
#pragma unroll
    for (int q = 0; q < V; ++q) rp2[q] = rm2[q] = (T)0;
    const bool nzr2 = in2 && (nz2 || nz1);
    if (in2 && !nzr2) VOF_STAT(8);
    if (nzr2) {
#pragma unroll
      for (int q = 0; q < V; ++q) fct_ratios<T>(c, t2[q], t3[q], tn[q], a2[q], a1[q], rp2[q], rm2[q]);
    }
    // stage C of face r-2: a minimum of two ratios, all of them zero unless one of the two cell rows has some
    T c2[V];
#pragma unroll
    for (int q = 0; q < V; ++q) c2[q] = (T)0;
    if (inc2 && (nzr2 || nzr3)) {
#pragma unroll
      for (int q = 0; q < V; ++q) c2[q] = fct_climit<T>(a2[q], rp3[q], rm3[q], rp2[q], rm2[q]);
    }
    // stage D of row r-3 (faces a3 below, a2 above)
    if (!(nz3 || nz2)) VOF_STAT(9);
    if (nz3 || nz2) {
#pragma unroll
      for (int q = 0; q < V; ++q) out[q] = fct_final<T, POST>(c, t3[q], a3[q], c3[q], a2[q], c2[q], d3[q]);
    } else {
#pragma unroll
      for (int q = 0; q < V; ++q) out[q] = fct_clamp<T, POST>(t3[q]);
    }
#pragma unroll
    for (int q = 0; q < V; ++q) {
      F1[q] = Fr[q]; u1[q] = ur[q]; L1[q] = Lr[q];
      a3[q] = a2[q]; a2[q] = a1[q]; a1[q] = ar[q];
      t3[q] = t2[q]; t2[q] = tn[q];
      d3[q] = d2[q]; d2[q] = dv1[q];
      rp3[q] = rp2[q]; rm3[q] = rm2[q]; c3[q] = c2[q];
    }
    nz3 = nz2; nz2 = nz1; nz1 = nz0; nzr3 = nzr2;
  }
};

// ------------------------------------------------------------------ one row of the fused transport, shared by k_transport and k_tm
// update_uv (2dvof.py:269-280) of the newest row and both FCT sweeps of solve_VOF_rudman (:312-318) as one iteration of a march
// along i: rows r of F, u*, v*, p come in, u and v of row r come out (through `on_uv`: a store, or k_tm's ring in LDS) and F'' of
// row r - 3.  The y sweep is row-local: a per-row stage in front of the x pipeline (YFIRST, even steps) or behind it.  Where the rows
// come from and go to, and which rows a chunk owns, is the caller's business -- one source for the arithmetic (round 6).
//   IN (an interior pair of k_tm): every row the march touches lies strictly inside [ilo, ihi] and every column in [2, ny].
template <typename T, int V>
struct TransportWindow {
  FctXPipe<T, V> pipe;
  T p1[V], rho1[V];        // p and rho of the previous row (update_uv's i-1 operands)
  int cls1;                // class of the previous row (see step); 2: rho1 holds the row's densities lane by lane
  T v1[V], v2[V], v3[V];   // x first: corrected v of rows r-1, r-2, r-3 (the y sweep trails the pipeline)

  // f1, pr1, v0: rows (first row of the march - 1) of F, p and v* -- the pipeline's first donor row.  YFIRST: that row enters the
  // pipeline y-swept (its corrected v needs operands of the same row only) unless it lies outside [ilo, ihi]
  template <bool YFIRST, bool IN>
  __device__ __forceinline__ void init(const Consts<T>& c, int j0, int ny, const T (&f1)[V], const T (&pr1)[V], const T (&v0)[V], bool swept) {
#pragma unroll
    for (int q = 0; q < V; ++q) { p1[q] = pr1[q]; rho1[q] = rho_of(c, f1[q]); }
    if (YFIRST && swept) {
      T vv[V], fs[V];
      const T rhol = lane_up(rho1[V - 1]), pl = lane_up(p1[V - 1]);
#pragma unroll
      for (int q = 0; q < V; ++q) {
        const int j = j0 + q;
        const T vn = corrected_velocity<T>(c, v0[q], rho1[q], q == 0 ? rhol : rho1[q - 1], p1[q],
                                           q == 0 ? pl : p1[q - 1], c.dyi);
        vv[q] = (IN || (j >= 2 && j <= ny)) ? vn : (T)0;
      }
      fct_y_row<T, V, false, IN>(c, j0, ny, f1, vv, fs);
      pipe.init(fs);
    } else {
      pipe.init(f1);
    }
    cls1 = 2;
#pragma unroll
    for (int q = 0; q < V; ++q) v1[q] = v2[q] = v3[q] = (T)0;
  }

  // Fr, ur, vr, pr: rows r of F, u*, v*, p (ur / vr become u / v of the row); out: F'' of row r - 3.  [ylo, yhi]: the rows whose
  // trailing y sweep (x first) this chunk needs.  on_uv(cls) is called when u and v of the row exist.
  template <bool YFIRST, bool IN, typename OnUV>
  __device__ __forceinline__ void step(const Consts<T>& c, int r, int ilo, int ihi, int nx, int j0, int ny, const T (&Fr)[V], T (&ur)[V],
                                       T (&vr)[V], const T (&pr)[V], T (&out)[V], int ylo, int yhi, OnUV&& on_uv) {
    // Class of the row as the wave sees it (wave-uniform): 0 = F is an exact 0 on every lane (gas), 1 = an exact 1
    // (liquid), 2 = anything else.  In a class-0 / class-1 row rho is rho_g / rho_l on every lane (rho_of(0) =
    // rho_g * 1 + rho_l * 0, rho_of(1) = rho_g * 0 + rho_l * 1: exact), and when the row below has the same class
    // update_uv's r = (rho + rho') / 2 is that density and dt / r the host's correctly rounded dt / rho.
    int cls = 2;
    {
      bool rz = true, ro = true;
#pragma unroll
      for (int q = 0; q < V; ++q) rz = rz && Fr[q] == (T)0;
      if (__all(rz)) {
        cls = 0;
      } else {
#pragma unroll
        for (int q = 0; q < V; ++q) ro = ro && Fr[q] == (T)1;
        if (__all(ro)) cls = 1;
      }
    }
    {  // update_uv for row r (:269-280): ur / vr hold u*[r] / v*[r]
      const T pl = lane_up(pr[V - 1]);
      const bool urow = IN || (r >= 2 && r <= nx);     // u exists on i in [2, nx]; the walls keep 0
      VOF_STAT(3);
      if (cls == 0) VOF_STAT(4);
      if (cls == 1) VOF_STAT(5);
      if (cls != 2 && cls == cls1) {
        VOF_STAT(6);
        const T k = cls ? c.dt_rho_l : c.dt_rho_g;
#pragma unroll
        for (int q = 0; q < V; ++q) {
          const int j = j0 + q;
          const T un = ur[q] - k * (pr[q] - p1[q]) * c.dxi;
          ur[q] = urow ? un : (T)0;
          const T vn = vr[q] - k * (pr[q] - (q == 0 ? pl : pr[q - 1])) * c.dyi;
          vr[q] = (IN || (j >= 2 && j <= ny)) ? vn : (T)0;
          p1[q] = pr[q];
        }
      } else {
        T rhor[V];
        if (cls1 != 2) {   // the row below went through the branch above: its densities are the constant
#pragma unroll
          for (int q = 0; q < V; ++q) rho1[q] = cls1 ? c.rho_l : c.rho_g;
        }
#pragma unroll
        for (int q = 0; q < V; ++q) rhor[q] = rho_of(c, Fr[q]);
        const T rhol = lane_up(rhor[V - 1]);
#pragma unroll
        for (int q = 0; q < V; ++q) {
          const int j = j0 + q;
          const T un = corrected_velocity<T>(c, ur[q], rhor[q], rho1[q], pr[q], p1[q], c.dxi);
          ur[q] = urow ? un : (T)0;
          const T vn = corrected_velocity<T>(c, vr[q], rhor[q], q == 0 ? rhol : rhor[q - 1], pr[q],
                                             q == 0 ? pl : pr[q - 1], c.dyi);
          vr[q] = (IN || (j >= 2 && j <= ny)) ? vn : (T)0;   // v exists on j in [2, ny]; j = 1, ny+1 keep set_BC's 0
          p1[q] = pr[q];
          rho1[q] = rhor[q];
        }
      }
      on_uv(urow);
      cls1 = cls;
    }
#pragma unroll
    for (int q = 0; q < V; ++q) out[q] = (T)0;
    const int io = r - 3;
    if (YFIRST) {
      // y sweep of row r in front of the pipeline; rows outside [ilo, ihi] (the ghost rows) enter
      // unswept, which is what the twin buffer holds for the x sweep in the two-kernel form
      T Fp[V];
      if ((!IN && (r < ilo || r > ihi)) || cls == 0) {
#pragma unroll
        for (int q = 0; q < V; ++q) Fp[q] = Fr[q];
      } else {
        fct_y_row<T, V, false, IN>(c, j0, ny, Fr, vr, Fp);
      }
      pipe.template push<true, IN>(c, r, ilo, ihi, Fp, ur, out, cls == 0);
    } else {
      T Fp[V];
      pipe.template push<false, IN>(c, r, ilo, ihi, Fr, ur, Fp, cls == 0);   // F'[r-3]
      if (io >= ylo && io <= yhi) {
        bool rz = true;
#pragma unroll
        for (int q = 0; q < V; ++q) rz = rz && Fp[q] == (T)0;
        if (!__all(rz)) fct_y_row<T, V, true, IN>(c, j0, ny, Fp, v3, out);
      }
#pragma unroll
      for (int q = 0; q < V; ++q) {
        v3[q] = v2[q]; v2[q] = v1[q]; v1[q] = vr[q];
      }
    }
  }
};

// CORR (full-domain handles only): the sweep that runs first also performs update_uv -- it
// computes u and v from u*, v*, p and F (rho) for the rows it streams, stores them, and feeds its
// own component straight into the flux pipeline.  `u` is then an output (Uo) and the wall faces
// i = 1, nx+1 carry the 0 that set_BC keeps there.
template <typename T, int V, bool POST, bool CORR>
__global__ __launch_bounds__(256) void k_fct_x(Geom g, Consts<T> c, const T* __restrict__ F,
                                                const T* __restrict__ u, T* __restrict__ Fn, int R,
                                                const T* __restrict__ us, const T* __restrict__ vs,
                                                const T* __restrict__ p, T* __restrict__ Uo, T* __restrict__ Vo,
                                                unsigned long long* __restrict__ courant, int rfirst, int rlast) {
  // rows [rfirst, rlast] (within [ilo, ihi]) are produced; the sweep's domain stays [ilo, ihi]
  WaveTimer wt_(WT_FCT_X);
  int j0, ra, rb;
  if (!wave_tile<V>(g, rfirst, rlast, R, j0, ra, rb)) return;
  const int ilo = g.ilo, ihi = g.ihi;
  FctXPipe<T, V> pipe;  // face -> Ftd -> rp/rm -> cx -> F' along i
  auto rowptr = [&](const T* base, int r) {
    int rc = r < g.row_lo ? g.row_lo : (r > g.row_hi ? g.row_hi : r);
    return base + at(g, rc, j0);
  };
  T F1[V];  // F[r-1]: the pipeline's first donor row, and update_uv's i-1 density
  load_c<T, V>(F1, rowptr(F, ra - 3));
  pipe.init(F1);
  T Fnx[V], unx[V];  // rows r of F and u (CORR: u*), prefetched one iteration ahead
  load_c<T, V>(Fnx, rowptr(F, ra - 2));
  load_c<T, V>(unx, rowptr(CORR ? us : u, ra - 2));
  // CORR state: p and rho of row r-1, prefetched p / v* / left neighbours of row r
  T p1[V], rho1[V];
  Row<T, V> pnx;
  T vsnx[V], Flnx = (T)0;
  unsigned int viol = 0;
  if (CORR) {
    load_c<T, V>(p1, rowptr(p, ra - 3));
#pragma unroll
    for (int q = 0; q < V; ++q) rho1[q] = rho_of(c, F1[q]);
    const T* pr0 = rowptr(p, ra - 2);
    load_c<T, V>(pnx.c, pr0);
    pnx.l = pr0[-1];
    load_s<T, V>(vsnx, rowptr(vs, ra - 2));
    Flnx = rowptr(F, ra - 2)[-1];
  }
  for (int r = ra - 2; r <= rb + 3; ++r) {
    T Fr[V], ur[V];
#pragma unroll
    for (int q = 0; q < V; ++q) {
      Fr[q] = Fnx[q];
      ur[q] = unx[q];
    }
    Row<T, V> pr;
    T vsr[V], Flr = Flnx;
    if (CORR) {
      pr = pnx;
#pragma unroll
      for (int q = 0; q < V; ++q) vsr[q] = vsnx[q];
    }
    if (r < rb + 3) {
      load_c<T, V>(Fnx, rowptr(F, r + 1));
      load_c<T, V>(unx, rowptr(CORR ? us : u, r + 1));
      if (CORR) {
        const T* prn = rowptr(p, r + 1);
        load_c<T, V>(pnx.c, prn);
        pnx.l = prn[-1];
        load_s<T, V>(vsnx, rowptr(vs, r + 1));
        Flnx = rowptr(F, r + 1)[-1];
      }
    }
    if (CORR) {  // update_uv for row r (:269-280): ur currently holds u*[r]
      T rhor[V], ov[V];
      const T rhol = rho_of(c, Flr);
      const bool urow = r >= 2 && r <= g.nx;   // u exists on i in [2, nx]; walls keep 0
      const bool own = r >= ra && r <= rb;     // rows this chunk stores (and counts)
#pragma unroll
      for (int q = 0; q < V; ++q) {
        const int j = j0 + q;
        rhor[q] = rho_of(c, Fr[q]);
        const T un = corrected_velocity<T>(c, ur[q], rhor[q], rho1[q], pr.c[q], p1[q], c.dxi);
        ur[q] = urow ? un : (T)0;
        const T rl = q == 0 ? rhol : rhor[q - 1];
        const T pl = q == 0 ? pr.l : pr.c[q - 1];
        const T vn = corrected_velocity<T>(c, vsr[q], rhor[q], rl, pr.c[q], pl, c.dyi);
        ov[q] = (j >= 2 && j <= g.ny) ? vn : (T)0;   // v exists on j in [2, ny]
        if (own && j <= g.ny && r >= g.own_lo && r <= g.own_hi) {
          if (urow && ur[q] * c.dt > c.cfl_x) viol++;
          if (j >= 2 && ov[q] * c.dt > c.cfl_y) viol++;
        }
      }
      if (own) {
        // the wall faces u[1], u[nx+1], v[:,1], v[:,ny+1] get set_BC's zeros (:525) here, because
        // the other sweep reads them before the u, v boundary kernel runs on a full domain
        store_s<T, V>(Uo + at(g, r, j0), ur, j0, 1, g.ny);
        store_s<T, V>(Vo + at(g, r, j0), ov, j0, 1, g.ny);
        if (j0 + V > g.ny) Vo[at(g, r, g.ny + 1)] = (T)0;
        if (r == g.nx) {
          T zero[V];
#pragma unroll
          for (int q = 0; q < V; ++q) zero[q] = (T)0;
          store_c<T, V>(Uo + at(g, r + 1, j0), zero, j0, 1, g.ny);
        }
      }
#pragma unroll
      for (int q = 0; q < V; ++q) {
        p1[q] = pr.c[q];
        rho1[q] = rhor[q];
      }
    }
    // Where F is identically 0 (the gas side of the interface) every flux, F~, limiter and the new
    // F are exact zeros: the pipeline bypasses itself once the wave's whole 7-row dependency window
    // is zero (FctXPipe::push).
    T out[V];
    bool rz = true;
#pragma unroll
    for (int q = 0; q < V; ++q) rz = rz && Fr[q] == (T)0;
    pipe.template push<POST>(c, r, ilo, ihi, Fr, ur, out, __all(rz));
    const int io = r - 3;
    if (io >= ra && io <= rb) store_s<T, V>(Fn + at(g, io, j0), out, j0, 1, g.ny);
  }
  if (CORR && __any(viol != 0)) {
    unsigned int tot = viol;
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) tot += __shfl_down(tot, sft, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(courant, (unsigned long long)tot);
  }
}
// 2dvof.py:385-448 fct_y_sweep, fused like k_fct_x.  The sweep direction is
// the contiguous one, so the +-3-cell dependency is resolved across lanes with
// wave shuffles: a wave owns 64*V consecutive cells of one row, of which the
// inner 64*V - 8 are valid outputs (tiles overlap by 8 columns; 4 keeps the
// 16-byte alignment of the lane accesses).  Rows are independent.
template <typename T, int V, bool POST, bool CORR>
__global__ __launch_bounds__(256) void k_fct_y(Geom g, Consts<T> c, const T* __restrict__ F,
                                                const T* __restrict__ v, T* __restrict__ Fn, int R, int nty,
                                                const T* __restrict__ us, const T* __restrict__ vs,
                                                const T* __restrict__ p, T* __restrict__ Uo, T* __restrict__ Vo,
                                                unsigned long long* __restrict__ courant, int rfirst, int rlast) {
  constexpr int W = 64 * V, HT = TileHalo::transport, STRIDE = W - 2 * HT;
  static_assert(HT >= 4 && HT % V == 0, "the y sweep's +-3 dependency is resolved across lanes");
  WaveTimer wt_(WT_FCT_Y);
  const int wave = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // SGPR: rows are wave-uniform
  const int lane = threadIdx.x & 63;
  const int tj = wave % nty, ch = wave / nty;
  const int c0 = 1 - HT + tj * STRIDE;
  const int j0 = c0 + lane * V;
  const int ra = rfirst + ch * R;
  if (ra > rlast) return;  // wave-uniform
  const int rb = ra + R - 1 < rlast ? ra + R - 1 : rlast;
  const int ny = g.ny;
  const int jlo = c0 + HT > 1 ? c0 + HT : 1;
  const int jhi = c0 + W - HT - 1 < ny ? c0 + W - HT - 1 : ny;
  size_t o = at(g, ra, j0);
  T Fnx[V], vnx[V];  // next row (CORR: v*), prefetched
  load_c<T, V>(Fnx, F + o);
  load_c<T, V>(vnx, (CORR ? vs : v) + o);
  // CORR (see k_fct_x): this sweep runs first and performs update_uv for its rows
  T p1[V], rho1[V], pnx[V], usnx[V];
  unsigned int viol = 0;
  if (CORR) {
    T f1[V];
    load_c<T, V>(p1, p + o - g.pitch);
    load_c<T, V>(f1, F + o - g.pitch);
#pragma unroll
    for (int q = 0; q < V; ++q) rho1[q] = rho_of(c, f1[q]);
    load_c<T, V>(pnx, p + o);
    load_s<T, V>(usnx, us + o);
  }
  for (int i = ra; i <= rb; ++i, o += g.pitch) {
    T Fz[V], vz[V], pz[V], usz[V];
#pragma unroll
    for (int q = 0; q < V; ++q) {
      Fz[q] = Fnx[q];
      vz[q] = vnx[q];
      if (CORR) {
        pz[q] = pnx[q];
        usz[q] = usnx[q];
      }
    }
    if (i < rb) {
      load_c<T, V>(Fnx, F + o + g.pitch);
      load_c<T, V>(vnx, (CORR ? vs : v) + o + g.pitch);
      if (CORR) {
        load_c<T, V>(pnx, p + o + g.pitch);
        load_s<T, V>(usnx, us + o + g.pitch);
      }
    }
    if (CORR) {  // update_uv for row i (:269-280): vz currently holds v*[i]
      T rhoz[V], ou[V];
#pragma unroll
      for (int q = 0; q < V; ++q) rhoz[q] = rho_of(c, Fz[q]);
      const T rhol = lane_up(rhoz[V - 1]), pl = lane_up(pz[V - 1]);
      const bool own = i >= g.own_lo && i <= g.own_hi;
#pragma unroll
      for (int q = 0; q < V; ++q) {
        const int j = j0 + q;
        const T un = corrected_velocity<T>(c, usz[q], rhoz[q], rho1[q], pz[q], p1[q], c.dxi);
        ou[q] = i >= 2 ? un : (T)0;                // u exists on i in [2, nx]
        const T vn = corrected_velocity<T>(c, vz[q], rhoz[q], q == 0 ? rhol : rhoz[q - 1], pz[q],
                                           q == 0 ? pl : pz[q - 1], c.dyi);
        vz[q] = (j >= 2 && j <= ny) ? vn : (T)0;   // v exists on j in [2, ny]; j = 1, ny+1 keep set_BC's 0
        if (own && j >= jlo && j <= jhi) {
          if (i >= 2 && ou[q] * c.dt > c.cfl_x) viol++;
          if (j >= 2 && vz[q] * c.dt > c.cfl_y) viol++;
        }
        p1[q] = pz[q];
        rho1[q] = rhoz[q];
      }
      // wall faces included (set_BC's zeros, :525): the x sweep reads u[1], u[nx+1] before the
      // u, v boundary kernel runs on a full domain
      store_s<T, V>(Uo + o, ou, j0, jlo, jhi);
      store_s<T, V>(Vo + o, vz, j0, jlo, jhi == ny ? ny + 1 : jhi);
      if (i == g.nx) {
        T zero[V];
#pragma unroll
        for (int q = 0; q < V; ++q) zero[q] = (T)0;
        store_c<T, V>(Uo + o + g.pitch, zero, j0, jlo, jhi);
      }
    }
    {  // F identically 0 over the wave's whole row segment: every output of the segment is 0
      bool rz = true;
#pragma unroll
      for (int q = 0; q < V; ++q) rz = rz && Fz[q] == (T)0;
      if (__all(rz)) {
        T zero[V];
#pragma unroll
        for (int q = 0; q < V; ++q) zero[q] = (T)0;
        store_s<T, V>(Fn + o, zero, j0, jlo, jhi);
        continue;
      }
    }
    T out[V];
    fct_y_row<T, V, POST>(c, j0, ny, Fz, vz, out);
    store_s<T, V>(Fn + o, out, j0, jlo, jhi);
  }
  if (CORR && __any(viol != 0)) {
    unsigned int tot = viol;
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) tot += __shfl_down(tot, sft, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(courant, (unsigned long long)tot);
  }
}

// ------------------------------------------------------------------ fused transport
// update_uv (2dvof.py:269-280) + BOTH FCT sweeps of solve_VOF_rudman (:312-318, :321-448) +
// post_process_f (:452-455) in one pass over F, u*, v*, p -> F'', u, v: the intermediate F' of the
// first sweep never goes to memory (7 array passes instead of 10 for the two-kernel form).
// Possible because the y sweep is row-local: while a wave marches along i for the x sweep's
// pipeline, the y sweep of a row is a per-row stage in front of that pipeline (YFIRST, even steps:
// y then x) or behind it (odd steps: x then y).  Same per-cell functions, same operands, same
// order as k_fct_x / k_fct_y, so F'' is identical.  Full domains only (the strip schedule ships u, v
// between the two sweeps).
//
// The reference applies no set_BC between the sweeps (S6): the second sweep sees F's ghost cells
// from before the first one.  Here those are simply the input's ghost cells: rows outside
// [ilo, ihi] enter the x pipeline unswept (YFIRST), and F' in the ghost columns only ever meets the
// zero wall velocity v[:,1] = v[:,ny+1] = 0 (x first).

// Up to three row ranges a launch produces, in this order, each cut in chunks of its own length
// (an empty range has last < first): e.g. the two edge bands of a strip in short chunks.
struct RowRanges {
  int first[3], last[3], R[3];
};

template <typename T, int V, bool YFIRST>
__global__ __launch_bounds__(256) void k_transport(Geom g, Consts<T> c, const T* __restrict__ F, T* __restrict__ Fn,
                                                    int nty, const T* __restrict__ us,
                                                    const T* __restrict__ vs, const T* __restrict__ p,
                                                    T* __restrict__ Uo, T* __restrict__ Vo,
                                                    unsigned long long* __restrict__ courant, RowRanges rr) {
  // rr: all computable rows of a full domain; on a strip the owned rows -- as one range, or the two
  // edge bands (what the neighbours wait for) first and then the rest, in one launch or in two.
  // The sweeps' domain stays [ilo, ihi].
  constexpr int W = 64 * V, HT = TileHalo::transport, STRIDE = W - 2 * HT;
  static_assert(HT >= 4 && HT % V == 0, "the y sweep's +-3 dependency is resolved across lanes");
  WaveTimer wt_(WT_TRANSPORT);
  const int wave = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int tj = wave % nty, ch = wave / nty;
  const int c0 = 1 - HT + tj * STRIDE;
  const int j0 = c0 + lane * V;
  const int ilo = g.ilo, ihi = g.ihi, nx = g.nx, ny = g.ny;
  int k = 0, cbase = 0;   // range of this chunk (wave-uniform)
  for (; k < 3; ++k) {
    const int n = rr.last[k] >= rr.first[k] ? (rr.last[k] - rr.first[k] + rr.R[k]) / rr.R[k] : 0;
    if (ch < cbase + n) break;
    cbase += n;
  }
  if (k == 3) return;  // padding waves of the last block
  const int R = rr.R[k], hi = rr.last[k];
  const int ra = rr.first[k] + (ch - cbase) * R;
  const int rb = ra + R - 1 < hi ? ra + R - 1 : hi;
  const int jlo = c0 + HT > 1 ? c0 + HT : 1;
  const int jhi = c0 + W - HT - 1 < ny ? c0 + W - HT - 1 : ny;
  auto rowptr = [&](const T* base, int r) {
    const int rc = r < g.row_lo ? g.row_lo : (r > g.row_hi ? g.row_hi : r);
    return base + at(g, rc, j0);
  };
  // the register window and the arithmetic of an iteration: TransportWindow (shared with k_tm)
  TransportWindow<T, V> win;
  {
    T f1[V], pr1[V], v0[V];
    load_c<T, V>(f1, rowptr(F, ra - 3));
    load_c<T, V>(pr1, rowptr(p, ra - 3));
    const bool swept = YFIRST && ra - 3 >= ilo;
    if (swept) {
      load_s<T, V>(v0, rowptr(vs, ra - 3));
    } else {
#pragma unroll
      for (int q = 0; q < V; ++q) v0[q] = (T)0;
    }
    win.template init<YFIRST, false>(c, j0, ny, f1, pr1, v0, swept);
  }
  T Fnx[V], usnx[V], vsnx[V], pnx[V];  // row r, prefetched one iteration ahead
  load_c<T, V>(Fnx, rowptr(F, ra - 2));
  load_s<T, V>(usnx, rowptr(us, ra - 2));
  if (YFIRST) {
    load_s<T, V>(vsnx, rowptr(vs, ra - 2));
  } else {
#pragma unroll
    for (int q = 0; q < V; ++q) vsnx[q] = (T)0;
  }
  load_c<T, V>(pnx, rowptr(p, ra - 2));
  unsigned int viol = 0;
  for (int r = ra - 2; r <= rb + 3; ++r) {
    T Fr[V], ur[V], vr[V], pr[V];
#pragma unroll
    for (int q = 0; q < V; ++q) {
      Fr[q] = Fnx[q]; ur[q] = usnx[q]; vr[q] = vsnx[q]; pr[q] = pnx[q];
    }
    if (r < rb + 3) {
      load_c<T, V>(Fnx, rowptr(F, r + 1));
      load_s<T, V>(usnx, rowptr(us, r + 1));
      // x first: the y sweep trails the pipeline and only touches the chunk's own rows, so v* of the
      // lead-in / lead-out rows is never used (the stored v of a row is written by the chunk that owns it)
      if (YFIRST || (r + 1 >= ra && r + 1 <= rb)) load_s<T, V>(vsnx, rowptr(vs, r + 1));
      load_c<T, V>(pnx, rowptr(p, r + 1));
    }
    T out[V];
    const bool own = r >= ra && r <= rb;     // rows this chunk stores (and counts)
    win.template step<YFIRST, false>(c, r, ilo, ihi, nx, j0, ny, Fr, ur, vr, pr, out, ra, rb, [&](bool urow) {
      if (own && r >= g.own_lo && r <= g.own_hi) {
#pragma unroll
        for (int q = 0; q < V; ++q) {
          const int j = j0 + q;
          if (j >= jlo && j <= jhi) {
            if (urow && ur[q] * c.dt > c.cfl_x) viol++;
            if (j >= 2 && vr[q] * c.dt > c.cfl_y) viol++;
          }
        }
      }
      if (own) {
        store_s<T, V>(Uo + at(g, r, j0), ur, j0, jlo, jhi);
        store_s<T, V>(Vo + at(g, r, j0), vr, j0, jlo, jhi == ny ? ny + 1 : jhi);
        if (r == nx) {
          T zero[V];
#pragma unroll
          for (int q = 0; q < V; ++q) zero[q] = (T)0;
          store_c<T, V>(Uo + at(g, r + 1, j0), zero, j0, jlo, jhi);
        }
      }
    });
    const int io = r - 3;
    if (io >= ra && io <= rb) store_s<T, V>(Fn + at(g, io, j0), out, j0, jlo, jhi);
  }
  if (__any(viol != 0)) {
    unsigned int tot = viol;
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) tot += __shfl_down(tot, sft, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(courant, (unsigned long long)tot);
  }
}

}  // namespace vof
