// kernels/jacobi.h -- k_jacobi (one sweep) and k_jacobi_tb (TS sweeps per launch), with the norm reductions of the residual-terminated solve
//
// Part of the gfx950 kernel set of the 2-D VOF hot path (see vof2d_kernels.h for the conventions:
// reference line citations, expression order, one wave = 64*V columns marching along i).
#pragma once
#include "common.h"
#include "plan.h"

namespace vof {

// ------------------------------------------------------------------ norms of a sweep (extension)
// max|p_new - p| and max|p_new| over the cells a wave stores: lane maxima -> wave maximum by
// __shfl_down -> one atomicMax per wave and norm on the bit pattern (non-negative doubles order like
// their bit patterns; +inf is the largest).  A NaN update counts as +inf, so a diverged field can
// never read as converged.
template <typename T>
__device__ __forceinline__ void norm_acc(T& upd, T& pmx, T pn, T po) {
  const T d = dabs<T>(pn - po), a = dabs<T>(pn);
  upd = d != d ? DivLimits<T>::inf : vmax(upd, d);
  pmx = a != a ? DivLimits<T>::inf : vmax(pmx, a);
}
template <typename T>
__device__ __forceinline__ void norm_publish(T upd, T pmx, unsigned long long* __restrict__ bits) {
  double r0 = (double)upd, r1 = (double)pmx;
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) {
    r0 = vmax(r0, __shfl_down(r0, s, 64));
    r1 = vmax(r1, __shfl_down(r1, s, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    if (r0 > 0.0) atomicMax(bits, (unsigned long long)__double_as_longlong(r0));
    if (r1 > 0.0) atomicMax(bits + 1, (unsigned long long)__double_as_longlong(r1));
  }
}

// ------------------------------------------------------------------ Jacobi
// 2dvof.py:258-266: one sweep p -> pn (ping-pong replaces the copy-back loop).
// North-star kernel: 3 arrays * sizeof(T) per cell of HBM traffic.  D rows of
// p and rhs are prefetched into registers ahead of use.  RESID additionally
// reduces max|pn - p| and max|pn| over owned rows (norm_acc / norm_publish above);
// not part of the reference (extension).
template <typename T, int V, int D, bool RESID>
__global__ __launch_bounds__(256) void k_jacobi(Geom g, Consts<T> c, const T* __restrict__ p,
                                                 const T* __restrict__ rhs, T* __restrict__ pn, int R,
                                                 unsigned long long* __restrict__ resid_bits) {
  WaveTimer wt_(WT_JACOBI);
  int j0, ra, rb;
  if (!wave_tile<V>(g, g.ilo, g.ihi, R, j0, ra, rb)) return;
  const int nx = g.nx, ny = g.ny;
  T an[V], as_[V], apI[V], yI[V];  // interior rows (ae = aw = dxi2): ap and its reciprocal per lane
#pragma unroll
  for (int q = 0; q < V; ++q) {
    an[q] = (j0 + q) != ny ? c.dyi2 : (T)0.0;
    as_[q] = (j0 + q) != 1 ? c.dyi2 : (T)0.0;
    apI[q] = (T)-1.0 * (c.dxi2 + c.dxi2 + an[q] + as_[q]);
    yI[q] = (T)1.0 / apI[q];
  }
  const int64_t pitch = g.pitch;
  size_t o = at(g, ra, j0);
  T w[V];
  Row<T, V> cur;
  load_c<T, V>(w, p + o - pitch);
  load_row<T, V>(cur, p + o);
  Row<T, V> qe[D];  // rows i+1 .. i+D of p
  T qb[D][V];       // rows i .. i+D-1 of rhs
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (ra + d <= rb) {
      load_row<T, V>(qe[d], p + o + (int64_t)(d + 1) * pitch);
      load_s<T, V>(qb[d], rhs + o + (int64_t)d * pitch);
    }
  }
  T res = (T)0, pmx = (T)0;
  for (int i0 = ra; i0 <= rb; i0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int i = i0 + d;
      if (i > rb) break;
      Row<T, V> e = qe[d];
      T b[V];
#pragma unroll
      for (int q = 0; q < V; ++q) b[q] = qb[d][q];
      if (i + D <= rb) {  // refill this slot with the rows D ahead
        load_row<T, V>(qe[d], p + o + (int64_t)(D + 1) * pitch);
        load_s<T, V>(qb[d], rhs + o + (int64_t)D * pitch);
      }
      const T ae = i != nx ? c.dxi2 : (T)0.0;
      const T aw = i != 1 ? c.dxi2 : (T)0.0;
      T out[V];
#pragma unroll
      for (int q = 0; q < V; ++q) {
        const T num = b[q] - ae * e.c[q] - aw * w[q] - an[q] * right_of(cur, q) - as_[q] * left_of(cur, q);
        if (i == 1 || i == nx) {  // wave-uniform: the wall rows have their own ap
          const T ap = (T)-1.0 * (ae + aw + an[q] + as_[q]);
          out[q] = div_by_const<T>(num, ap, (T)1.0 / ap);   // (the hardware a / b double-rounds subnormal ties)
        } else {
          out[q] = div_by_const<T>(num, apI[q], yI[q]);
        }
        if (RESID) {
          if (i >= g.own_lo && i <= g.own_hi && j0 + q <= ny) norm_acc<T>(res, pmx, out[q], cur.c[q]);
        }
      }
      store_s<T, V>(pn + o, out, j0, 1, ny);
#pragma unroll
      for (int q = 0; q < V; ++q) w[q] = cur.c[q];
      cur = e;
      o += pitch;
    }
  }
  if (RESID) norm_publish<T>(res, pmx, resid_bits);
}

// ------------------------------------------------------------------ Jacobi, TS sweeps per launch
// Temporal blocking of 2dvof.py:258-266.  The reference runs a fixed number of sweeps (10, :521)
// with a sweep-invariant rhs, so TS consecutive sweeps can be applied while a tile streams through
// registers once: stage s (= sweep s of this launch) trails stage s-1 by one row.  HBM traffic per
// launch stays 3 arrays (read p, read rhs, write p_TS) for TS sweeps.  Each cell value is computed
// by the same expression, in the same order, from the same operands as TS single sweeps, so the
// result is identical.  A wave owns 64*V columns; intermediate sweeps exchange their j+-1
// neighbours across lanes (DPP), which costs TS-1 invalid columns on each tile side (tiles
// overlap by 2*H, H = TS-1 rounded up to V) and TS rows of lead-in/lead-out per chunk.
//
// Register rotation: stage s keeps rows i-1, i, i+1 of its input in a ring of three row buffers
// whose roles advance by one per iteration, and the rhs rows in a ring of six; the row loop is
// unrolled by 6 with compile-time ring positions, so no value is ever moved between registers.
// RESID (extension): the LAST of the TS sweeps also reduces max|p_TS - p_(TS-1)| and
// max|p_TS| over the owned cells the tile stores, so the residual-terminated solve keeps the TS-sweep
// fusion.  p_(TS-1) of a row is the previous stage's output one iteration earlier (kept in V extra
// registers when SQ, where the ring holds products rather than values).
// BS (even ny, fields below 2 GiB; chosen by the launch wrapper): unconditional loads and a range-checked buffer store
// (store_buf_nt), so that the compiler's waits are exact counts -- see k_momentum.
template <typename T, int V, int TS, bool SQ, bool RESID = false, bool BS = false>
__global__ __launch_bounds__(256) void k_jacobi_tb(Geom g, Consts<T> c, const T* __restrict__ p,
                                                    const T* __restrict__ rhs, T* __restrict__ pn, int R,
                                                    int ntt, unsigned long long* __restrict__ norm_bits = nullptr,
                                                    TbPlan tp = TbPlan{nullptr, nullptr, 0, 0, 0, 0, 0}, int first = 1, int last = 0) {
  // rows [first, last] of the result are produced (last < first: all of [g.ilo, g.ihi])
  if (last < first) { first = g.ilo; last = g.ihi; }
  // SQ (dxi2 == dyi2 bitwise, i.e. square cells): the stencil has ONE off-diagonal coefficient, so
  // the product coef * p[i,j] is the same number in the equations of all four neighbours of (i,j).
  // Stages 2.. then receive products instead of values -- 1 multiply per cell-sweep instead of 4
  // -- and their numerator is b - cE - cW - cN - cS in the reference's order with bit-identical
  // terms.  The zero coefficients of the walls (:258-261) are reproduced at the producer: rows
  // outside [1, nx] publish 0 * value, and cells in columns outside [1, ny] carry the value 0
  // (their reciprocal yI is 0, so div_by_const returns 0), whose product is the same exact zero.
  static_assert(TS >= 2 && TS <= 5, "rhs ring holds 6 rows");
  WaveTimer wt_(WT_JACOBI_TB);
  constexpr int W = 64 * V;
  // invalid columns per tile side after TS sweeps: TS-1 from the cross-lane exchange of sweeps
  // 2..TS, plus 1 when the first sweep also takes its j-neighbours from adjacent lanes (SQ)
  constexpr int H = ((TS - 1 + (SQ ? 1 : 0) + V - 1) / V) * V;
  constexpr int STRIDE = W - 2 * H;
  int lblock = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int wave = lblock * (blockDim.x >> 6) + wib;
  const int lane = threadIdx.x & 63;
  // wave -> (tile column tj, rows [ra, rb]): chunks of R rows of every column, or, while the tiny-value
  // front crosses the grid, the equal-cost chunks of the step's plan (tb_make_plan)
  int tj = 0, ra = 1, rb = 0;
  bool planned = false;
  if (tp.masks != nullptr) {
    const unsigned long long* pl = tp.plan;
    if (pl[0] == plan_key(tp)) {   // an active plan of this launch's own geometry
      const unsigned long long e = pl[1 + wave];    // scalar loads: nobody writes the plan during the launch
      tj = (int)(e & 0xffull);
      ra = (int)((e >> 8) & 0xfffffffull);
      rb = (int)((e >> 36) & 0xfffffffull);
      planned = true;
    }
  }
  if (!planned) {
    // A launch sized for the whole grid's plan that produces part of the rows (enqueue_steps_halves) and finds no plan:
    // the blocks the uniform layout needs are spread evenly over the launch's blocks, so that every XCD gets its
    // share (the first blocks alone would all sit on half of the XCDs).
    const int need = ((((last - first) / R + 1) * ntt) + 3) >> 2, have = (int)gridDim.x;
    if (need < have) {
      const int lo = (int)(((long long)lblock * need) / have), hi = (int)(((long long)(lblock + 1) * need) / have);
      if (hi == lo) return;
      wave = lo * (blockDim.x >> 6) + wib;
    }
    tj = wave % ntt; ra = first + (wave / ntt) * R; rb = ra + R - 1;
  }
  tj = __builtin_amdgcn_readfirstlane(tj);
  ra = __builtin_amdgcn_readfirstlane(ra);
  rb = __builtin_amdgcn_readfirstlane(rb);
  if (planned && rb < ra) return;   // an unused wave of the plan
  if (planned && ra < first) ra = first;   // (a launch on part of the rows takes its part of every planned chunk)
  const int c0 = 1 - H + tj * STRIDE;
  const int j0 = c0 + lane * V;
  if (ra > last) return;  // wave-uniform
  if (rb > last) rb = last;
  if (rb < ra) return;
  const int nx = g.nx, ny = g.ny;
  const int jlo = c0 + H > 1 ? c0 + H : 1;
  const int jhi = c0 + W - H - 1 < ny ? c0 + W - H - 1 : ny;
  const int64_t pitch = g.pitch;
  int hit = 0;   // this lane ran the tiny-numerator tier (adaptive layout: reported per tile column)

  T an[V], as_[V], apI[V], yI[V];
#pragma unroll
  for (int q = 0; q < V; ++q) {
    const int j = j0 + q;
    an[q] = j != ny ? c.dyi2 : (T)0.0;
    as_[q] = j != 1 ? c.dyi2 : (T)0.0;
    apI[q] = (T)-1.0 * (c.dxi2 + c.dxi2 + an[q] + as_[q]);  // ap of rows 1 < i < nx
    yI[q] = (T)1 / apI[q];
    if (SQ && (j < 1 || j > ny)) yI[q] = (T)0;  // out-of-domain columns: every sweep yields the value 0
  }
  auto rowptr = [&](const T* base, int r) {
    const int rc = r < g.row_lo ? g.row_lo : (r > g.row_hi ? g.row_hi : r);
    return base + (size_t)(rc - g.row_lo) * (size_t)pitch + (size_t)(g.col0 + j0);
  };

  // ring[s][k]: input rows of stage s+1 (s = 0: values of p from memory; s > 0: the previous
  // stage's output, as products when SQ).  In the sub-iteration with phase U (t = tb + U):
  //   ring[s][(U+0)%3] = row i-1,  ring[s][(U+1)%3] = row i,  ring[s][(U+2)%3] = row i+1 (incoming)
  // where i = t - (s+1).
  T ring[TS][3][V];
  T sideL[3], sideR[3];  // general: j0-1 / j0+V of the memory rows (unused when SQ: DPP instead)
  T rq[6][V];  // rhs row x lives in slot (x - (t0-1)) mod 6
#pragma unroll
  for (int s = 0; s < TS; ++s)
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int q = 0; q < V; ++q) ring[s][k][q] = (T)0;
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int q = 0; q < V; ++q) rq[k][q] = (T)0;

  T pv[V], pv_new[V];        // RESID && SQ: values of sweep TS-1, row i (pv) / row i+1 (pv_new) of the last stage
  T upd = (T)0, pmx = (T)0;  // RESID: lane maxima of |p_TS - p_(TS-1)| and |p_TS|
#pragma unroll
  for (int q = 0; q < V; ++q) pv[q] = pv_new[q] = (T)0;

  const int t0 = ra - TS + 2, t1 = rb + TS;
  // phase 0 at t = t0: rows t0-2, t0-1, t0 of p in ring[0][0..2]; rhs row t0-1 in rq slot 0
  // (stage s at phase U reads rhs row t-s -> slot (U+1-s) mod 6)
  load_c<T, V>(ring[0][0], rowptr(p, t0 - 2));
  {
    const T* q1 = rowptr(p, t0 - 1);
    load_c<T, V>(ring[0][1], q1);
    const T* q2 = rowptr(p, t0);
    load_c<T, V>(ring[0][2], q2);
    if constexpr (!SQ) {
      sideL[0] = sideR[0] = (T)0;
      sideL[1] = q1[-1];
      sideR[1] = q1[V];
      sideL[2] = q2[-1];
      sideR[2] = q2[V];
    }
  }
  load_s<T, V>(rq[0], rowptr(rhs, t0 - 1));

  const T* const pn_tile = pn + (int64_t)(g.col0 + c0);   // BS: (wave-uniform) the tile's first column of stored row row_lo
  const int voff_st = (j0 >= jlo && j0 + V - 1 <= jhi) ? lane * (int)(V * sizeof(T)) : kBufSkip;
  auto sub = [&](auto uc, int t) {
    constexpr int U = decltype(uc)::value;
    constexpr int kM = U % 3, kC = (U + 1) % 3, kE = (U + 2) % 3;
    T carry[V];  // output (a value of p) of the previous stage = row i+1 of this stage's input
#pragma unroll
    for (int s = 1; s <= TS; ++s) {
      const int i = t - s;
      T sl, sr;
      if (s == 1 && !SQ) {  // (SQ: sideL/sideR are never touched)
        sl = sideL[kC];
        sr = sideR[kC];
      } else {
        sl = lane_up_z(ring[s - 1][kC][V - 1]);   // (tile edge lanes: columns in the overlap, recomputed next door)
        sr = lane_dn_z(ring[s - 1][kC][0]);
      }
      if (s > 1) {
        if (SQ) {  // publish the previous stage's row i+1 as products (zero coefficient outside [1, nx])
          const T coef = (i + 1 >= 1 && i + 1 <= nx) ? c.dxi2 : (T)0.0;
#pragma unroll
          for (int q = 0; q < V; ++q) ring[s - 1][kE][q] = coef * carry[q];
        } else {
#pragma unroll
          for (int q = 0; q < V; ++q) ring[s - 1][kE][q] = carry[q];
        }
      }
      // lead-in: stage s first matters at row ra-(TS-s), i.e. from t = ra-TS+2s on (wave-uniform)
      if (s > 1 && t < ra - TS + 2 * s) continue;
      // Rows outside [ilo, ihi] and columns outside [1, ny] are computed like any other cell:
      // their values are finite and only ever enter a valid cell multiplied by a zero
      // coefficient (aw/ae at the walls, as/an at j = 1 / ny) or sit in the invalid fringe.
      const bool edge = (i == 1) || (i == nx);
      const T ae = i != nx ? c.dxi2 : (T)0.0;
      const T aw = i != 1 ? c.dxi2 : (T)0.0;
      const int slot = ((U + 1 - s) % 6 + 6) % 6;  // constant after unrolling
      T num[V];
#pragma unroll
      for (int q = 0; q < V; ++q) {
        const T N = q == V - 1 ? sr : ring[s - 1][kC][q + 1];
        const T S = q == 0 ? sl : ring[s - 1][kC][q - 1];
        if (SQ && s > 1)  // inputs of stages 2.. are products; stage 1 reads values of p from memory
          num[q] = rq[slot][q] - ring[s - 1][kE][q] - ring[s - 1][kM][q] - N - S;
        else
          num[q] = rq[slot][q] - ae * ring[s - 1][kE][q] - aw * ring[s - 1][kM][q] - an[q] * N - as_[q] * S;
      }
      if (edge) {  // wave-uniform: first / last interior row has its own ap
#pragma unroll
        for (int q = 0; q < V; ++q) {
          const T ap = (T)-1.0 * (ae + aw + an[q] + as_[q]);
          T o = div_by_const<T>(num[q], ap, (T)1.0 / ap);     // (the hardware a / b double-rounds subnormal ties)
          if (SQ && ((j0 + q) < 1 || (j0 + q) > ny)) o = (T)0;  // same zero the interior rows produce
          carry[q] = o;
        }
      } else {
        div_by_const_v<T, V>(carry, num, apI, yI, &hit);
      }
      if constexpr (RESID) {
        if (SQ && s == TS - 1) {
#pragma unroll
          for (int q = 0; q < V; ++q) pv_new[q] = carry[q];
        }
        if (s == TS && i >= ra && i <= rb && i >= g.own_lo && i <= g.own_hi) {
#pragma unroll
          for (int q = 0; q < V; ++q)
            if (j0 + q >= jlo && j0 + q <= jhi) norm_acc<T>(upd, pmx, carry[q], SQ ? pv[q] : ring[TS - 1][kC][q]);
        }
      }
      if (s == 1 && (BS || t < t1)) {
        // ring[0][kM] (row t-2) is dead now: prefetch row t+1 into it; rhs row t into the free slot
        const T* qn = rowptr(p, t + 1);
        load_c<T, V>(ring[0][kM], qn);
        if constexpr (!SQ) {
          sideL[kM] = qn[-1];
          sideR[kM] = qn[V];
        }
        load_s<T, V>(rq[(U + 1) % 6], rowptr(rhs, t));
      }
    }
    const int io = t - TS;
    if constexpr (BS)
      store_buf_nt<T, V>(pn_tile, (io >= ra && io <= rb) ? voff_st : kBufSkip,
                         (io >= ra && io <= rb) ? (int)((int64_t)(io - g.row_lo) * pitch * (int64_t)sizeof(T)) : 0, carry);   // (a dropped row keeps an in-field offset)
    else if (io >= ra && io <= rb)
      store_s<T, V>(pn + at(g, io, j0), carry, j0, jlo, jhi);
    if constexpr (RESID && SQ) {
#pragma unroll
      for (int q = 0; q < V; ++q) pv[q] = pv_new[q];
    }
  };

  for (int t = t0; t <= t1; t += 6) {
    sub(IC<0>{}, t);
    if (t + 1 > t1) break;
    sub(IC<1>{}, t + 1);
    if (t + 2 > t1) break;
    sub(IC<2>{}, t + 2);
    if (t + 3 > t1) break;
    sub(IC<3>{}, t + 3);
    if (t + 4 > t1) break;
    sub(IC<4>{}, t + 4);
    if (t + 5 > t1) break;
    sub(IC<5>{}, t + 5);
  }
  if constexpr (RESID) norm_publish<T>(upd, pmx, norm_bits);
  if (tp.masks != nullptr && __any(hit != 0) && lane == 0) {   // report the (row band, tile column) cells of this chunk
    for (int b = tb_band_of(g, ra); b <= tb_band_of(g, rb); ++b) atomicOr(tp.masks + tb_word(tp.par ^ 1, b, tj), 1ull << (tj & 63));
  }
}

}  // namespace vof
