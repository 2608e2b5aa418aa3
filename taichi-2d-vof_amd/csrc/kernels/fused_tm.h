// kernels/fused_tm.h -- k_tm: k_transport of step n and k_momentum of step n + 1 as ONE kernel
//
// Part of the gfx950 kernel set of the 2-D VOF hot path (see vof2d_kernels.h for the conventions).
//
// k_transport ends a step by writing F'', u, v; k_momentum begins the next by reading exactly those three arrays:
// six array passes (of the step's 19) that only carry data from one launch to the next.  Here a workgroup is a PAIR
// of waves on one tile: wave 0 runs the transport march of k_transport, wave 1 the momentum march of k_momentum five
// rows behind it, and the rows of F'', u, v go from the one to the other through an 8-row ring in LDS -- the
// momentum wave issues no global load at all.  F'' is still stored (the next transport reads it), u and v only when
// the caller wants them in memory (STORE_UV: the last step of a batch; nothing inside a batch reads them).
//   reads  F, u*, v*, p (4 passes)      writes F'', u*', v*', rhs' (4 passes) [+ u, v]        instead of 7 + 6
// Same per-cell functions, same operands, same order as the two kernels: the values are theirs.
//   * tile: 128 columns; the transport march is valid on [c0 + 4, c0 + 123] (the y sweep resolves its +-3 dependency
//     across lanes), the momentum march on what it can form from those (it reaches 3 columns to either side):
//     [c0 + 7, c0 + 120]; HF = 8 keeps a lane's first column odd, i.e. its 16-byte loads aligned: tiles advance by
//     112 columns (7 cache lines: every stored segment starts and ends on a line), everything is stored on
//     [c0 + 8, c0 + 119];
//   * rows: a chunk [ma, mb] of momentum output needs F'' rows ma-3 .. mb+3 and u, v rows ma-2 .. mb+2, so the
//     transport wave marches t = ma-5 .. mb+6 (producing u[t], v[t], F''[t-3]) and the momentum wave runs its
//     iteration r' = t - 5 at step t: it reads F''[r'] (and F''[1] for the mirrored ghost row 0), produced at
//     step <= r' + 4, and u / v row r' - 1.  One barrier per step; ring slot = row & 7;
//   * u*', v*' go to a second pair of arrays (chunks of other workgroups still read the old u*, v*); the caller
//     alternates the pairs.
// Virtual ghosts (the steady-state fused step) only; on a strip the rows beyond its stored rows are the clamped edge rows (the invalid fringe).
// The two marches below are k_transport's and k_momentum's row loops with the source / sink of F'', u, v exchanged (LDS
// instead of memory) and the chunk bounds of the pair; they are kept as copies, not shared with the stand-alone kernels,
// so that those kernels' register allocation and instruction schedule stay what profiles/ measured.
#pragma once
#include "momentum.h"
#include "transport.h"

namespace vof {

struct TmGeom { static constexpr int HF = 8; };   // invalid columns per tile side of the fused march: 4 (transport) + 3 (momentum), rounded up to even

template <typename T, int V>
struct TmRing {
  static constexpr int W = 64 * V, NR = 8;
  T f[NR][W], u[NR][W], v[NR][W];
};

template <typename T, int V>
__device__ __forceinline__ void ring_put(T (&row)[64 * V], int lane, const T (&c)[V]) {
  Pack<T, V> k;
#pragma unroll
  for (int q = 0; q < V; ++q) k.v[q] = c[q];
  *reinterpret_cast<Pack<T, V>*>(&row[lane * V]) = k;
}
template <typename T, int V>
__device__ __forceinline__ void ring_get(Row<T, V>& w, const T (&row)[64 * V], int lane) {
  const Pack<T, V> k = *reinterpret_cast<const Pack<T, V>*>(&row[lane * V]);
#pragma unroll
  for (int q = 0; q < V; ++q) w.c[q] = k.v[q];
  w.l = row[lane * V - (lane > 0 ? 1 : 0)];                   // (tile edge lanes: columns in the invalid fringe)
  w.r = row[lane * V + V - (lane < 63 ? 0 : 1)];
}

// IN (interior pair, chosen per workgroup): every row the pair touches lies in [3, nx - 1] and every column of its tile in
// [2, ny] -- no wall row, no ghost column, no clamp.  The marches are the same code with the wall tests folded to
// constants: the row / column selects around every result (okN, okK, okP, urow, dom ...), the ghost mirrors, the
// clamped 64-bit row addresses and most of the scalar bookkeeping go (a third of a wave's instructions, in a kernel whose
// waves issue one instruction at a time and wait for each other at every row); 93 % of the pairs of a 4096^2 grid.

// ------------------------------------------------------------------ transport march (k_transport's, rows tra .. trb)
template <typename T, int V, bool YFIRST, bool STORE_UV, bool BS, bool IN, int ABL>
__device__ __forceinline__ void tm_transport_march(const Geom& g, const Consts<T>& c, TmRing<T, V>& ring, const T* __restrict__ F,
                                                   T* __restrict__ Fn, const T* __restrict__ us, const T* __restrict__ vs,
                                                   const T* __restrict__ p, T* __restrict__ Uo, T* __restrict__ Vo,
                                                   unsigned long long* __restrict__ courant, int c0, int lane, int ma, int mb,
                                                   WaveTimer& wt_) {
  constexpr int W = 64 * V, HF = TmGeom::HF;
  // the transport wave is the one the momentum wave waits for: priority 1 (4096^2 fp64: 314 -> 299 us y first, 327 -> 305 x first;
  // the momentum wave instead: 314 / 333).  ABL_PRIO0 of the diagnostic build = without it.
  if constexpr ((ABL & ABL_PRIO0) == 0) __builtin_amdgcn_s_setprio(1);
  const int j0 = c0 + lane * V;
  const int ilo = g.ilo, ihi = g.ihi, nx = g.nx, ny = g.ny;
  const int jlo = IN ? c0 + HF : (c0 + HF > 1 ? c0 + HF : 1);
  const int jhi = IN ? c0 + W - HF - 1 : (c0 + W - HF - 1 < ny ? c0 + W - HF - 1 : ny);
  const int t_lo = ma - 5, t_hi = mb + 8;   // lockstep steps of the pair
  const int64_t pitch = g.pitch;
  auto rowptr = [&](const T* base, int r) {
    if constexpr ((ABL & ABL_FIXED_ROW) != 0) r = ma;
    const int rc = IN ? r : (r < g.row_lo ? g.row_lo : (r > g.row_hi ? g.row_hi : r));
    return base + at(g, rc, j0);
  };
  const int tra = ma - 3, trb = mb + 3;
  FctXPipe<T, V> pipe;
  T p1[V], rho1[V];
  {
    T f1[V];
    load_c<T, V>(f1, rowptr(F, tra - 3));
    load_c<T, V>(p1, rowptr(p, tra - 3));
#pragma unroll
    for (int q = 0; q < V; ++q) rho1[q] = rho_of(c, f1[q]);
    if (YFIRST && (IN || tra - 3 >= ilo)) {
      T v0[V], fs[V];
      load_s<T, V>(v0, rowptr(vs, tra - 3));
      const T rhol = lane_up(rho1[V - 1]), pl = lane_up(p1[V - 1]);
#pragma unroll
      for (int q = 0; q < V; ++q) {
        const int j = j0 + q;
        const T vn = corrected_velocity<T>(c, v0[q], rho1[q], q == 0 ? rhol : rho1[q - 1], p1[q],
                                           q == 0 ? pl : p1[q - 1], c.dyi);
        v0[q] = (IN || (j >= 2 && j <= ny)) ? vn : (T)0;
      }
      fct_y_row<T, V, false, IN>(c, j0, ny, f1, v0, fs);
      pipe.init(fs);
    } else {
      pipe.init(f1);
    }
  }
  int cls1 = 2;
  T v1[V], v2[V], v3[V];
#pragma unroll
  for (int q = 0; q < V; ++q) v1[q] = v2[q] = v3[q] = (T)0;
  T Fnx[V], usnx[V], vsnx[V], pnx[V];
  load_c<T, V>(Fnx, rowptr(F, tra - 2));
  load_s<T, V>(usnx, rowptr(us, tra - 2));
  load_s<T, V>(vsnx, rowptr(vs, tra - 2));
  load_c<T, V>(pnx, rowptr(p, tra - 2));
  // BS: the one store of the march (F'') is a range-checked buffer store and the loads run one row past the march (a row
  // that exists), so every memory instruction is unconditional and the wait for the rows requested a step ago is an
  // exact count that leaves the store in flight
  const T* const Fn_tile = Fn + (int64_t)(g.col0 + c0);
  const int voff_st = (j0 >= jlo && j0 + V - 1 <= jhi) ? lane * (int)(V * sizeof(T)) : kBufSkip;
  unsigned int viol = 0;
  for (int t = t_lo; t <= t_hi; ++t) {
    if (t <= trb + 3) {
      const int r = t;
      T Fr[V], ur[V], vr[V], pr[V];
#pragma unroll
      for (int q = 0; q < V; ++q) {
        Fr[q] = Fnx[q]; ur[q] = usnx[q]; vr[q] = vsnx[q]; pr[q] = pnx[q];
      }
      if (BS || r < trb + 3) {
        load_c<T, V>(Fnx, rowptr(F, r + 1));
        load_s<T, V>(usnx, rowptr(us, r + 1));
        load_s<T, V>(vsnx, rowptr(vs, r + 1));
        load_c<T, V>(pnx, rowptr(p, r + 1));
      }
      if constexpr ((ABL & ABL_PASS0) != 0) {   // timing only: the loaded rows go on as they are
#pragma unroll
        for (int q = 0; q < V; ++q) Fr[q] += pr[q] * (T)0;
        ring_put<T, V>(ring.u[r & 7], lane, ur);
        ring_put<T, V>(ring.v[r & 7], lane, vr);
        ring_put<T, V>(ring.f[(r - 3) & 7], lane, Fr);
        if (r - 3 >= ma && r - 3 <= mb && !(ABL & ABL_NO_STORE)) store_s<T, V>(Fn + at(g, r - 3, j0), Fr, j0, jlo, jhi);
        wt_.barrier();
        continue;
      }
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
      {  // update_uv for row r (:269-280)
        const T pl = lane_up(pr[V - 1]);
        const bool urow = IN || (r >= 2 && r <= nx);
        const bool own = r >= ma && r <= mb;
        if (cls != 2 && cls == cls1) {
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
          if (cls1 != 2) {
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
            vr[q] = (IN || (j >= 2 && j <= ny)) ? vn : (T)0;
            p1[q] = pr[q];
            rho1[q] = rhor[q];
          }
        }
        if (own && (IN || (r >= g.own_lo && r <= g.own_hi))) {
#pragma unroll
          for (int q = 0; q < V; ++q) {
            const int j = j0 + q;
            if (j >= jlo && j <= jhi) {
              if (urow && ur[q] * c.dt > c.cfl_x) viol++;
              if ((IN || j >= 2) && vr[q] * c.dt > c.cfl_y) viol++;
            }
          }
        }
        ring_put<T, V>(ring.u[r & 7], lane, ur);
        ring_put<T, V>(ring.v[r & 7], lane, vr);
        if constexpr (STORE_UV && BS && IN && !(ABL & ABL_NO_STORE)) {
          // (an interior pair: no wall row, no wall column -- u and v leave like F'', range-checked buffer stores)
          const int so = own ? (int)((int64_t)(r - g.row_lo) * pitch * (int64_t)sizeof(T)) : 0;
          store_buf_nt<T, V>(Uo + (int64_t)(g.col0 + c0), own ? voff_st : kBufSkip, so, ur);
          store_buf_nt<T, V>(Vo + (int64_t)(g.col0 + c0), own ? voff_st : kBufSkip, so, vr);
        } else if (STORE_UV && own && !(ABL & ABL_NO_STORE)) {
          store_s<T, V>(Uo + at(g, r, j0), ur, j0, jlo, jhi);
          store_s<T, V>(Vo + at(g, r, j0), vr, j0, jlo, jhi == ny ? ny + 1 : jhi);
          if (r == nx) {
            T zero[V];
#pragma unroll
            for (int q = 0; q < V; ++q) zero[q] = (T)0;
            store_c<T, V>(Uo + at(g, r + 1, j0), zero, j0, jlo, jhi);
          }
        }
        cls1 = cls;
      }
      T out[V];
#pragma unroll
      for (int q = 0; q < V; ++q) out[q] = (T)0;
      const int io = r - 3;
      if (YFIRST) {
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
        if (io >= tra && io <= trb) {
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
      ring_put<T, V>(ring.f[io & 7], lane, out);
      const bool st = io >= ma && io <= mb;
      if constexpr ((ABL & ABL_NO_STORE) != 0) {
        asm volatile("" :: "v"(out[0]), "v"(out[V - 1]));
      } else if constexpr (BS) {
        store_buf_nt<T, V>(Fn_tile, st ? voff_st : kBufSkip, st ? (int)((int64_t)(io - g.row_lo) * pitch * (int64_t)sizeof(T)) : 0, out);
      } else if (st) {
        store_s<T, V>(Fn + at(g, io, j0), out, j0, jlo, jhi);
      }
    }
    wt_.barrier();
  }
  if (__any(viol != 0)) {
    unsigned int tot = viol;
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) tot += __shfl_down(tot, sft, 64);
    if (lane == 0) atomicAdd(courant, (unsigned long long)tot);
  }
}

// -------------------------------------------------------------------- momentum march (k_momentum's, rows ma .. mb)
template <typename T, int V, bool BS, bool IN, int ABL>
__device__ __forceinline__ void tm_momentum_march(const Geom& g, const Consts<T>& c, TmRing<T, V>& ring, T* __restrict__ us_out,
                                                  T* __restrict__ vs_out, T* __restrict__ rhs, int c0, int lane, int ma, int mb,
                                                  WaveTimer& wt_) {
  constexpr int W = 64 * V, HF = TmGeom::HF;
  if constexpr ((ABL & ABL_PRIO1) != 0) __builtin_amdgcn_s_setprio(1);
  const int t_lo = ma - 5, t_hi = mb + 8;   // lockstep steps of the pair
  if constexpr ((ABL & ABL_IDLE1) != 0) {
    for (int t = t_lo; t <= t_hi; ++t) wt_.barrier();
    return;
  }
  const int j0 = c0 + lane * V;
  const int ilo = g.ilo, ihi = g.ihi, nx = g.nx, ny = g.ny;
  const int jlo = IN ? c0 + HF : (c0 + HF > 1 ? c0 + HF : 1);
  const int jhi = IN ? c0 + W - HF - 1 : (c0 + W - HF - 1 < ny ? c0 + W - HF - 1 : ny);
  const T dt = c.dt, dxi = c.dxi, dyi = c.dyi, dxi2 = c.dxi2, dyi2 = c.dyi2;
  bool dom[V];
#pragma unroll
  for (int q = 0; q < V; ++q) dom[q] = IN || ((j0 + q) >= 1 && (j0 + q) <= ny);
  auto mirrow = [&](int r) { return IN ? r : (r == 0 ? 1 : (r == nx + 1 ? nx : r)); };   // virtual ghost rows of F and v (:176-189)
  const bool edge_cols = !IN && (c0 - 1 <= 0 || c0 + W >= ny + 1);
  auto get_F = [&](Row<T, V>& w, int r) { ring_get<T, V>(w, ring.f[mirrow(r) & 7], lane); };
  auto get_u = [&](Row<T, V>& w, int r) { ring_get<T, V>(w, ring.u[r & 7], lane); };
  auto get_v = [&](Row<T, V>& w, int r) { ring_get<T, V>(w, ring.v[mirrow(r) & 7], lane); };
  Row<T, V> F2, F1;
  T F3c[V];
  Row<T, V> u3, u2, v3, v2;
  T mx2[V], mx3[V], my2[V];
  T k3[V];
  T us3[V], vs3[V];
  T rho3[V];
  const int r0 = ma - 1, r1 = mb + 3;
  const T* const us_tile = us_out + (int64_t)(g.col0 + c0);
  const T* const vs_tile = vs_out + (int64_t)(g.col0 + c0);
  const T* const rhs_tile = rhs + (int64_t)(g.col0 + c0);
  const int voff_st = (j0 >= jlo && j0 + V - 1 <= jhi) ? lane * (int)(V * sizeof(T)) : kBufSkip;
  auto zero_row = [](Row<T, V>& w) {
    w.l = w.r = (T)0;
#pragma unroll
    for (int q = 0; q < V; ++q) w.c[q] = (T)0;
  };
  zero_row(u3); zero_row(u2); zero_row(v3); zero_row(v2);
  zero_row(F2); zero_row(F1);
#pragma unroll
  for (int q = 0; q < V; ++q) F3c[q] = mx2[q] = mx3[q] = my2[q] = k3[q] = us3[q] = vs3[q] = rho3[q] = (T)0;
  bool flat2 = true, flat1 = true, flat0;
  // Rows of uniform F (gas, or liquid away from the interface: nine rows in ten of a dam-break) -- wave-level history of
  // `flat` (the three newest rows uniform and equal), bit k = the iteration k before this one.  What it lets a row skip is
  // exact: the skipped arithmetic would produce the same zeros / the same rho, nu from the same F (round 6).
  unsigned flat_hist = 0u;
  // (the force terms of a flat window are +-0; they enter u*, v* as (... + gx) + fx: the sum in front is never -0 unless gx is)
  const bool zero_force_ok = !(c.gx == (T)0 && __builtin_signbit(c.gx)) && !(c.gy == (T)0 && __builtin_signbit(c.gy));
  Row<T, V> un, vn;   // u / v row r-1 of the coming iteration
  zero_row(un); zero_row(vn);
  for (int t = t_lo; t <= t_hi; ++t) {
    const int r = t - 5;
    if (r == r0) {   // the window's first two rows (k_momentum loads them in front of its loop)
      get_F(F2, r0 - 2);
      get_F(F1, r0 - 1);
      if (edge_cols) {
        mirror_ghost_cols<T, V>(F2, j0, ny);
        mirror_ghost_cols<T, V>(F1, j0, ny);
      }
      flat2 = row_flat<T, V>(F2);
      flat1 = row_flat<T, V>(F1);
    }
    if (r >= r0 && r <= r1) {
      Row<T, V> F0, u1 = un;
      const Row<T, V> v1 = vn;
      get_F(F0, r);
      get_u(un, r);          // (rows r of u, v: used by the next iteration, as in k_momentum)
      get_v(vn, r);
      if (edge_cols) {
        mirror_ghost_cols<T, V>(F0, j0, ny);
        mirror_ghost_cols<T, V>(u1, j0, ny);
      }
      // ---- N: normals of row r-1 (:285-306)
      const bool okN = IN || ((r - 1) >= ilo && (r - 1) <= ihi);
      T mx1[V], my1[V];
      flat0 = row_flat<T, V>(F0);
      const bool flat = flat2 && flat1 && flat0 && F2.c[0] == F1.c[0] && F1.c[0] == F0.c[0];
      const bool wflat = __all(flat);
#ifdef VOF_NO_FLAT_SHORTCUTS      // (A/B builds: make variant NAME=noflat EXTRA=-DVOF_NO_FLAT_SHORTCUTS)
      flat_hist = 0u;
#else
      flat_hist = __builtin_amdgcn_readfirstlane((flat_hist << 1) | (wflat ? 1u : 0u));
#endif
      if (wflat) {
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
      T us2[V], vs2[V], rho2[V], nu2[V];
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
          T dudx = upwind_diff<T>(u00 > 0, u00, um0, up0) * dxi;
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
        us2[q] = (okP && (IN || i >= 2) && dom[q]) ? ou : (T)0;
        vs2[q] = (okP && (IN || (j >= 2 && j <= ny))) ? ov : (T)0;
      }
      if constexpr ((ABL & ABL_NO_STORE) != 0) {
        asm volatile("" :: "v"(us2[0]), "v"(us2[V - 1]), "v"(vs2[0]), "v"(vs2[V - 1]));
      } else if constexpr (BS) {
        const bool rowok = i >= ma && i <= mb;
        const int vo = rowok ? voff_st : kBufSkip;
        const int so = rowok ? (int)((int64_t)(i - g.row_lo) * g.pitch * (int64_t)sizeof(T)) : 0;
        store_buf_nt<T, V>(us_tile, vo, so, us2);
        store_buf_nt<T, V>(vs_tile, vo, so, vs2);
      } else if (i >= ma && i <= mb) {
        if (i >= 2) store_s<T, V>(us_out + at(g, i, j0), us2, j0, jlo, jhi);
        store_s<T, V>(vs_out + at(g, i, j0), vs2, j0, jlo > 2 ? jlo : 2, jhi);
      }
      // ---- R: rhs of row r-3 (:239-241)
      const int i3 = r - 3;
      if (BS || (i3 >= ma && i3 <= mb)) {
        const T vsr = lane_dn(vs3[0]);
        T out[V];
#pragma unroll
        for (int q = 0; q < V; ++q) {
          const T vright = q == V - 1 ? vsr : vs3[q + 1];
          out[q] = div_by_const_inrange<T>(rho3[q], c.dt, c.inv_dt) *
                   ((us2[q] - us3[q]) * c.dxi + (vright - vs3[q]) * c.dyi);
        }
        if constexpr ((ABL & ABL_NO_STORE) != 0)
          asm volatile("" :: "v"(out[0]), "v"(out[V - 1]));
        else if constexpr (BS)
          store_buf_nt<T, V>(rhs_tile, (i3 >= ma && i3 <= mb) ? voff_st : kBufSkip,
                             (i3 >= ma && i3 <= mb) ? (int)((int64_t)(i3 - g.row_lo) * g.pitch * (int64_t)sizeof(T)) : 0, out);
        else
          store_s<T, V>(rhs + at(g, i3, j0), out, j0, jlo, jhi);
      }
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
    wt_.barrier();
  }
}

// (three waves per SIMD: <= 168 VGPRs in every instantiation -- the forms with exec-masked global stores would take 171)
template <typename T, int V, bool YFIRST, bool STORE_UV, bool BS, int ABL = 0>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(3))) void k_tm(Geom g, Consts<T> c, const T* __restrict__ F, T* __restrict__ Fn, int ntf,
                                            const T* __restrict__ us, const T* __restrict__ vs, const T* __restrict__ p,
                                            T* __restrict__ Uo, T* __restrict__ Vo, T* __restrict__ us_out,
                                            T* __restrict__ vs_out, T* __restrict__ rhs,
                                            unsigned long long* __restrict__ courant, int R, TbPlan tp, int first, int last,
                                            int first2 = 1, int last2 = 0) {
  // rows [first, last] and -- the two edge bands of a strip in one launch -- [first2, last2] (last2 < first2: none), each cut in chunks of R
  constexpr int W = 64 * V, HF = TmGeom::HF, STRIDE = W - 2 * HF;
  static_assert(HF >= 4 + 3 && HF % V == 0, "momentum's inputs must lie inside the transport march's valid columns");
  static_assert(sizeof(TbPlanShared) <= sizeof(TmRing<double, 2>) / 2, "the planner block borrows the ring's LDS");
  __shared__ __attribute__((aligned(16))) char smem[sizeof(TmRing<T, V>) > sizeof(TbPlanShared) ? sizeof(TmRing<T, V>) : sizeof(TbPlanShared)];
  const int plan_blocks = tp.masks != nullptr ? 1 : 0;
  if (plan_blocks && blockIdx.x == 0) {   // the planner of the next step's k_jacobi_tb launches, as in k_momentum
    tb_make_plan(g, tp, *reinterpret_cast<TbPlanShared*>(smem));
    return;
  }
  TmRing<T, V>& ring = *reinterpret_cast<TmRing<T, V>*>(smem);
  WaveTimer wt_(WT_TM);
  int pair = (int)blockIdx.x - plan_blocks;
  // Workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md "Workgroup dispatch"; observed, not promised: only speed depends on
  // it) and each XCD has its own L2.  Within a window of 8 * kXcdGroup consecutive workgroups -- about two chunk rows, in
  // flight together -- XCD x takes kXcdGroup ADJACENT tiles, whose 16 overlapping columns it then finds in its own L2
  // instead of fetching them a second time: k_tm 279 -> 270 us (4096^2, in the step), with the same in k_jacobi_pair
  // 0.4375 -> 0.430 ms/step (profiles/r05_xcd_groups_ab.txt).  Unlike xcd_contiguous_block the dispatch order stays one
  // compact band.
  pair = xcd_grouped_block(pair, (int)gridDim.x - plan_blocks);
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0: transport, 1: momentum
  const int lane = threadIdx.x & 63;
  const int tj = pair % ntf, ch = pair / ntf;
  const int c0 = 1 - HF + tj * STRIDE;
  int ma = first + ch * R, lim = last;
  if (ma > last) {         // (block-uniform)
    const int n1 = last >= first ? (last - first + R) / R : 0;
    ma = first2 + (ch - n1) * R;
    lim = last2;
    if (last2 < first2 || ma > last2) return;   // both waves leave
  }
  const int mb = ma + R - 1 < lim ? ma + R - 1 : lim;
  // interior pair, stated as what the IN marches fold to constants: every row the pair loads, forms or tests (ma - 8 .. mb + 7: the
  // x pipeline's "row strictly inside [ilo, ihi]" tests reach ma - 7 and mb + 6) lies inside the computable rows [ilo, ihi] -- on a
  // full domain [1, nx], on a strip also inside the stored rows, whose addresses the interior marches do not clamp --, every row it
  // counts Courant violations on is an owned row, columns c0 - 1 .. c0 + W lie in [1, ny + 1] with every lane's columns in [2, ny]
  const bool interior = ma - 8 >= g.ilo && mb + 7 <= g.ihi && ma >= g.own_lo && mb <= g.own_hi && c0 >= 2 && c0 + W - 1 <= g.ny;
  if (role == 0) {
    if (interior) tm_transport_march<T, V, YFIRST, STORE_UV, BS, true, ABL>(g, c, ring, F, Fn, us, vs, p, Uo, Vo, courant, c0, lane, ma, mb, wt_);
    else tm_transport_march<T, V, YFIRST, STORE_UV, BS, false, ABL>(g, c, ring, F, Fn, us, vs, p, Uo, Vo, courant, c0, lane, ma, mb, wt_);
  } else {
    if (interior) tm_momentum_march<T, V, BS, true, ABL>(g, c, ring, us_out, vs_out, rhs, c0, lane, ma, mb, wt_);
    else tm_momentum_march<T, V, BS, false, ABL>(g, c, ring, us_out, vs_out, rhs, c0, lane, ma, mb, wt_);
  }
}

}  // namespace vof
