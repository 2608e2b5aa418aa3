// kernels/jacobi_pair.h -- k_jacobi_pair: 2 * TS Jacobi sweeps in one launch, by pairs of waves
//
// Part of the gfx950 kernel set of the 2-D VOF hot path (see vof2d_kernels.h for the conventions).
//
// Two k_jacobi_tb<TS> launches read p and rhs twice and write p twice.  Here a workgroup is a pair of waves on one tile:
// wave 0 runs k_jacobi_tb's march (TS sweeps, square cells) on the chunk widened by TS rows per side and hands its
// result rows -- and the rhs rows it has loaded -- to wave 1 through two 8-row rings in LDS; wave 1 runs the same
// march TS + 2 rows behind it, taking its "memory" rows from the rings, and stores the result of all 2 * TS sweeps:
// 3 array passes per 2 * TS sweeps instead of 6.  The register pipeline of each wave is k_jacobi_tb's (a single wave
// doing 2 * TS stages needs 250 VGPRs, or an rhs ring in LDS and twice the serial work per row: both measured slower).
// Same stage arithmetic, same operands, same order: the values are those of two k_jacobi_tb launches.
//   * tile: 128 columns; each march loses TS columns per side (TS - 1 cross-lane sweeps + the first sweep's DPP
//     neighbours): H = 10, tiles advance by 108 columns;
//   * rows: the pair produces rows [ra, rb]; the first march produces [ra - TS, rb + TS] from p rows [ra - 2 TS, rb + 2 TS];
//   * lockstep, one barrier per step: at step tau the first wave runs its sub-iteration t = tau (writing its result
//     row t - TS and the rhs row t - 1 into the rings), the second its sub-iteration t = tau - (TS + 2), which reads
//     result row t + 1 (written at step t + TS + 1) and rhs row t (written at step t + 1); ring slot = row & 7.
// Square cells only (the product-carrying pipeline); the caller keeps k_jacobi_tb for everything else.
#pragma once
#include "jacobi.h"

namespace vof {

template <typename T, int V>
struct JpRing {
  static constexpr int W = 64 * V, NR = 8;
  T p[NR][W], b[NR][W];
  int hit;   // a lane of the pair ran the tiny-numerator tier (the report to the next step's work plan)
};
template <typename T, int V>
__device__ __forceinline__ void jp_put(T (&row)[64 * V], int lane, const T (&c)[V]) {
  Pack<T, V> k;
#pragma unroll
  for (int q = 0; q < V; ++q) k.v[q] = c[q];
  *reinterpret_cast<Pack<T, V>*>(&row[lane * V]) = k;
}
template <typename T, int V>
__device__ __forceinline__ void jp_get(T (&c)[V], const T (&row)[64 * V], int lane) {
  const Pack<T, V> k = *reinterpret_cast<const Pack<T, V>*>(&row[lane * V]);
#pragma unroll
  for (int q = 0; q < V; ++q) c[q] = k.v[q];
}

// ROLE 0: the first TS sweeps (memory -> rings); ROLE 1: the second TS sweeps (rings -> memory).  [ra, rb]: the rows
// THIS march produces.
template <typename T, int V, int TS, bool BS, int ROLE, int ABL = 0>
__device__ __forceinline__ void jacobi_pair_march(const Geom& g, const Consts<T>& c, const T* __restrict__ p,
                                                  const T* __restrict__ rhs, T* __restrict__ pn, JpRing<T, V>& lds,
                                                  int c0, int lane, int ra, int rb, WaveTimer& wt) {
  // The first wave of a pair is the one the second waits for at every barrier (it also loads, and writes two rows to
  // LDS): it runs at priority 1 -- 4096^2 fp64: 146 -> 135 us per launch inside the front, 149 -> 141 behind it
  // (tools/probes/pair_bound.py; the other way round: 142 / 148).  ABL_PRIO0 of the diagnostic build = without it.
  if constexpr ((ROLE == 0 && !(ABL & ABL_PRIO0)) || (ROLE == 1 && (ABL & ABL_PRIO1))) __builtin_amdgcn_s_setprio(1);
  constexpr int W = 64 * V;
  constexpr int H = ((2 * TS + V - 1) / V) * V;   // TS invalid columns per side and march (TS - 1 cross-lane sweeps + the first sweep's DPP neighbours)
  const int j0 = c0 + lane * V;
  const int nx = g.nx, ny = g.ny;
  const int jlo = c0 + H > 1 ? c0 + H : 1;
  const int jhi = c0 + W - H - 1 < ny ? c0 + W - H - 1 : ny;
  const int64_t pitch = g.pitch;
  // Every stage takes PRODUCTS coef * p, the first one too: a row that arrives (from memory / from the first march) is
  // multiplied once by the coefficient it carries in the equations of all its four neighbours -- dxi2 (= dyi2: square
  // cells), or 0 where the reference's ae / aw / an / as is zero, i.e. for rows outside [1, nx] and columns outside
  // [1, ny] -- instead of four times, once per neighbour (2dvof.py:258-263: the same factors, the same products).
  T cj[V], apI[V], yI[V];
#pragma unroll
  for (int q = 0; q < V; ++q) {
    const int j = j0 + q;
    const T an = j != ny ? c.dyi2 : (T)0.0, as_ = j != 1 ? c.dyi2 : (T)0.0;
    cj[q] = (j >= 1 && j <= ny) ? c.dxi2 : (T)0.0;
    apI[q] = (T)-1.0 * (c.dxi2 + c.dxi2 + an + as_);  // ap of rows 1 < i < nx
    yI[q] = (T)1 / apI[q];
    if (j < 1 || j > ny) yI[q] = (T)0;  // out-of-domain columns: every sweep yields the value 0
  }
  auto rowptr = [&](const T* base, int r) {
    if constexpr ((ABL & ABL_FIXED_ROW) != 0) r = ra;
    const int rc = r < g.row_lo ? g.row_lo : (r > g.row_hi ? g.row_hi : r);
    return base + (size_t)(rc - g.row_lo) * (size_t)pitch + (size_t)(g.col0 + j0);
  };
  auto to_products = [&](T (&row)[V], int r) {
    const bool rowok = r >= 1 && r <= nx;
#pragma unroll
    for (int q = 0; q < V; ++q) row[q] = (rowok ? cj[q] : (T)0.0) * row[q];
  };
  T ring[TS][3][V];
  T rq[6][V];
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
  const int t0 = ra - TS + 2, t1 = rb + TS;
  if constexpr (ROLE == 0) {
    load_c<T, V>(ring[0][0], rowptr(p, t0 - 2));
    load_c<T, V>(ring[0][1], rowptr(p, t0 - 1));
    load_c<T, V>(ring[0][2], rowptr(p, t0));
    load_s<T, V>(rq[0], rowptr(rhs, t0 - 1));
  } else {
    for (int k = 0; k < 2 * TS + 2; ++k) wt.barrier();   // the steps the first march is ahead
    jp_get<T, V>(ring[0][0], lds.p[(t0 - 2) & 7], lane);
    jp_get<T, V>(ring[0][1], lds.p[(t0 - 1) & 7], lane);
    jp_get<T, V>(ring[0][2], lds.p[t0 & 7], lane);
    jp_get<T, V>(rq[0], lds.b[(t0 - 1) & 7], lane);
  }
  to_products(ring[0][0], t0 - 2);
  to_products(ring[0][1], t0 - 1);   // (row t0 in the first sub-iteration, like every row after it)
  const T* const pn_tile = pn + (int64_t)(g.col0 + c0);
  const int voff_st = (j0 >= jlo && j0 + V - 1 <= jhi) ? lane * (int)(V * sizeof(T)) : kBufSkip;
  // FAST sub-iterations (wave-uniform, taken in groups of six): every stage is past its lead-in, none of the rows
  // t - TS .. t touches a wall row (all in [2, nx - 1]: every coefficient is dxi2, ap = apI), the rows to prefetch exist,
  // the output row t - TS is produced.  The wall / lead-in / clamp tests of the general form -- as many scalar
  // instructions as vector ones, in a kernel whose waves issue one instruction at a time, the second wave of the pair
  // waiting for the first at every row -- and the 64-bit row-address products are gone: running addresses serve the loads
  // and the store.
  // (buffer loads and stores: BS -- even ny, fields below 2 GiB; other handles keep the general form throughout)
  const T* const p_tile = p + (int64_t)(g.col0 + c0);
  const T* const rhs_tile = rhs + ((int64_t)(g.col0 + c0) - pitch);   // + the offset of row t + 1 = rhs row t
  const int voff_ld = lane * (int)(V * sizeof(T));
  int soff = 0;          // FAST: byte offset of row t + 1 (first march: the rows to request) / of the output row t - TS (second march)
  auto sub = [&](auto uc, auto fc, int t) {
    constexpr int U = decltype(uc)::value;
    constexpr bool FAST = decltype(fc)::value != 0;
    constexpr int kM = U % 3, kC = (U + 1) % 3, kE = (U + 2) % 3;
    T carry[V];
    if constexpr (ROLE == 1 && (ABL & ABL_IDLE1)) { wt.barrier(); return; }
    if constexpr (ROLE == 0) jp_put<T, V>(lds.b[(t - 1) & 7], lane, rq[U % 6]);   // rhs row t - 1 (loaded an iteration ago) for the second march
    if constexpr (ROLE == 0 && (ABL & ABL_PASS0)) {
      const int io = t - TS;
      if (io >= ra && io <= rb) jp_put<T, V>(lds.p[io & 7], lane, ring[0][kC]);
      if (t < t1) {
        load_c<T, V>(ring[0][kM], rowptr(p, t + 1));
        load_s<T, V>(rq[(U + 1) % 6], rowptr(rhs, t));
      }
      wt.barrier();
      return;
    }
    // row t of the march's input, requested in the previous sub-iteration
    if constexpr (FAST) {
#pragma unroll
      for (int q = 0; q < V; ++q) ring[0][kE][q] = cj[q] * ring[0][kE][q];
    } else {
      to_products(ring[0][kE], t);
    }
#pragma unroll
    for (int s = 1; s <= TS; ++s) {
      const int i = t - s;
      const T sl = lane_up_z(ring[s - 1][kC][V - 1]);
      const T sr = lane_dn_z(ring[s - 1][kC][0]);
      if (s > 1) {   // the previous stage's row i + 1 as products (zero coefficient outside [1, nx])
        const T coef = (FAST || (i + 1 >= 1 && i + 1 <= nx)) ? c.dxi2 : (T)0.0;
#pragma unroll
        for (int q = 0; q < V; ++q) ring[s - 1][kE][q] = coef * carry[q];
      }
      if (!FAST && s > 1 && t < ra - TS + 2 * s) continue;
      const bool edge = !FAST && ((i == 1) || (i == nx));
      const int slot = ((U + 1 - s) % 6 + 6) % 6;
      T num[V];
#pragma unroll
      for (int q = 0; q < V; ++q) {
        const T N = q == V - 1 ? sr : ring[s - 1][kC][q + 1];
        const T S = q == 0 ? sl : ring[s - 1][kC][q - 1];
        num[q] = rq[slot][q] - ring[s - 1][kE][q] - ring[s - 1][kM][q] - N - S;
      }
      if (edge) {   // wave-uniform: the first / last interior row has its own ap
        const T ae = i != nx ? c.dxi2 : (T)0.0;
        const T aw = i != 1 ? c.dxi2 : (T)0.0;
#pragma unroll
        for (int q = 0; q < V; ++q) {
          int j = j0 + q;
          asm volatile("" : "+v"(j));   // (or an / as of the two wall rows are hoisted out of the row loop: eight registers for 4096 rows)
          const T an = j != ny ? c.dyi2 : (T)0.0, as_ = j != 1 ? c.dyi2 : (T)0.0;
          const T ap = (T)-1.0 * (ae + aw + an + as_);
          T o = div_by_const<T>(num[q], ap, (T)1.0 / ap);
          if (j < 1 || j > ny) o = (T)0;
          carry[q] = o;
        }
      } else {
        div_by_const_v<T, V, false, ColdFlagLds>(carry, num, apI, yI, ColdFlagLds{&lds.hit});
      }
      if (s == 1) {
        if constexpr (FAST) {
          if constexpr (ROLE == 0) {
            load_buf<T, V>(ring[0][kM], p_tile, voff_ld, soff, false);
            load_buf<T, V>(rq[(U + 1) % 6], rhs_tile, voff_ld, soff, true);
            if constexpr ((ABL & ABL_FIXED_ROW) == 0) soff += (int)(pitch * (int64_t)sizeof(T));
          } else {
            jp_get<T, V>(ring[0][kM], lds.p[(t + 1) & 7], lane);
            jp_get<T, V>(rq[(U + 1) % 6], lds.b[t & 7], lane);
          }
        } else if (t < t1) {
          if constexpr (ROLE == 0) {
            load_c<T, V>(ring[0][kM], rowptr(p, t + 1));
            load_s<T, V>(rq[(U + 1) % 6], rowptr(rhs, t));
          } else {
            jp_get<T, V>(ring[0][kM], lds.p[(t + 1) & 7], lane);
            jp_get<T, V>(rq[(U + 1) % 6], lds.b[t & 7], lane);
          }
        }
      }
    }
    const int io = t - TS;
    if constexpr (ROLE == 0) {
      if (FAST || (io >= ra && io <= rb)) jp_put<T, V>(lds.p[io & 7], lane, carry);
    } else if constexpr ((ABL & ABL_NO_STORE) != 0) {
      asm volatile("" :: "v"(carry[0]), "v"(carry[V - 1]));
    } else if constexpr (BS) {
      if constexpr (FAST) {
        store_buf_nt<T, V>(pn_tile, voff_st, soff, carry);
        soff += (int)(pitch * (int64_t)sizeof(T));
      } else {
        store_buf_nt<T, V>(pn_tile, (io >= ra && io <= rb) ? voff_st : kBufSkip,
                           (io >= ra && io <= rb) ? (int)((int64_t)(io - g.row_lo) * pitch * (int64_t)sizeof(T)) : 0, carry);
      }
    } else if (FAST || (io >= ra && io <= rb)) {
      store_s<T, V>(pn + at(g, io, j0), carry, j0, jlo, jhi);
    }
    wt.barrier();
  };
  // The row loop runs in three phases: general sub-iterations while the stages start up, FAST ones in groups of six
  // (rows t - TS .. t + 5 in [2, nx - 1], every stage past its lead-in, rows up to t + 6 stored, t + 5 < t1), general
  // ones to the end.  (Three loops, not one loop with two bodies: the two bodies would meet in the rings' phi nodes.)
  auto imax3 = [](int a, int b, int d) { return a > b ? (a > d ? a : d) : (b > d ? b : d); };
  auto imin3 = [](int a, int b, int d) { return a < b ? (a < d ? a : d) : (b < d ? b : d); };
  const int fast_lo = imax3(ra + TS, TS + 2, g.row_lo + TS);     // first t
  const int fast_hi = imin3(nx - 1, t1 - 1, g.row_hi - 1);       // last t
  int t = t0;
  bool done = false;   // a general group ended at t1
  for (int phase = 0; phase < 2; ++phase) {
    for (; !done && t <= t1 && (phase == 1 || t < fast_lo || t + 5 > fast_hi); t += 6) {
      sub(IC<0>{}, IC<0>{}, t);
      if (t + 1 > t1) { done = true; break; }
      sub(IC<1>{}, IC<0>{}, t + 1);
      if (t + 2 > t1) { done = true; break; }
      sub(IC<2>{}, IC<0>{}, t + 2);
      if (t + 3 > t1) { done = true; break; }
      sub(IC<3>{}, IC<0>{}, t + 3);
      if (t + 4 > t1) { done = true; break; }
      sub(IC<4>{}, IC<0>{}, t + 4);
      if (t + 5 > t1) { done = true; break; }
      sub(IC<5>{}, IC<0>{}, t + 5);
    }
    if (BS && phase == 0 && !done && t <= t1 && t >= fast_lo && t + 5 <= fast_hi) {
      soff = (int)((int64_t)((ROLE == 0 ? t + 1 : t - TS) - g.row_lo) * pitch * (int64_t)sizeof(T));
      for (; t + 5 <= fast_hi; t += 6) {
        sub(IC<0>{}, IC<1>{}, t);
        sub(IC<1>{}, IC<1>{}, t + 1);
        sub(IC<2>{}, IC<1>{}, t + 2);
        sub(IC<3>{}, IC<1>{}, t + 3);
        sub(IC<4>{}, IC<1>{}, t + 4);
        sub(IC<5>{}, IC<1>{}, t + 5);
      }
    }
  }
  // Both waves of the pair arrive at the same number of barriers (rb - ra + 4 * TS + 1 with the pair's [ra, rb]): the first
  // march runs 2 * TS rows more than the second, which starts 2 * TS + 2 barriers late.
  if constexpr (ROLE == 0) { wt.barrier(); wt.barrier(); }
}

template <typename T, int V, int TS, bool BS, int ABL = 0>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4))) void k_jacobi_pair(Geom g, Consts<T> c, const T* __restrict__ p,
                                                     const T* __restrict__ rhs, T* __restrict__ pn, int R, int ntt,
                                                     TbPlan tp, int first, int last) {
  constexpr int W = 64 * V;
  constexpr int H = ((2 * TS + V - 1) / V) * V;   // TS invalid columns per side and march (TS - 1 cross-lane sweeps + the first sweep's DPP neighbours)
  constexpr int STRIDE = W - 2 * H;
  __shared__ __attribute__((aligned(16))) JpRing<T, V> lds;
  WaveTimer wt_(WT_JACOBI_PAIR);
  if (last < first) { first = g.ilo; last = g.ihi; }
  const int pair = (int)blockIdx.x;
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  // pair -> (tile column tj, rows [ra, rb]): chunks of R rows of every column, or the equal-cost chunks of the step's plan
  int upair;          // the uniform layout's index
  upair = xcd_grouped_block(pair, (int)gridDim.x);   // (XCD x takes adjacent tiles of a window: see k_tm)
  int tj = upair % ntt, ra = first + (upair / ntt) * R, rb = ra + R - 1;
  bool planned = false;
  if (tp.masks != nullptr) {
    const unsigned long long* pl = tp.plan;
    if (pl[0] == plan_key(tp)) {   // an active plan of this launch's own geometry
      const unsigned long long e = pl[1 + pair];
      tj = (int)(e & 0xffull);
      ra = (int)((e >> 8) & 0xfffffffull);
      rb = (int)((e >> 36) & 0xfffffffull);
      planned = true;
    }
  }
  tj = __builtin_amdgcn_readfirstlane(tj);
  ra = __builtin_amdgcn_readfirstlane(ra);
  rb = __builtin_amdgcn_readfirstlane(rb);
  if (planned && rb < ra) return;   // (block-uniform: both waves of the pair leave)
  if (planned && ra < first) ra = first;
  if (ra > last) return;
  if (rb > last) rb = last;
  if (rb < ra) return;
  const int c0 = 1 - H + tj * STRIDE;
  if (threadIdx.x == 0) lds.hit = 0;   // (the first barrier comes before the second wave's first sub-iteration, and the first wave's rows are rows again)
  __syncthreads();
  if (role == 0) {
    jacobi_pair_march<T, V, TS, BS, 0, ABL>(g, c, p, rhs, pn, lds, c0, lane, ra - TS, rb + TS, wt_);
  } else {
    jacobi_pair_march<T, V, TS, BS, 1, ABL>(g, c, p, rhs, pn, lds, c0, lane, ra, rb, wt_);
  }
  if (tp.masks != nullptr) {   // report the (row band, tile column) cells of this chunk
    __syncthreads();
    if (threadIdx.x == 0 && lds.hit != 0) {
      for (int b = tb_band_of(g, ra < g.ilo ? g.ilo : ra); b <= tb_band_of(g, rb > g.ihi ? g.ihi : rb); ++b)
        atomicOr(tp.masks + tb_word(tp.par ^ 1, b, tj), 1ull << (tj & 63));
    }
  }
}

}  // namespace vof
