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
template <typename T, int V, int TS, bool BS, int ROLE>
__device__ __forceinline__ void jacobi_pair_march(const Geom& g, const Consts<T>& c, const T* __restrict__ p,
                                                  const T* __restrict__ rhs, T* __restrict__ pn, JpRing<T, V>& lds,
                                                  int c0, int lane, int ra, int rb, int& hit) {
  constexpr int W = 64 * V;
  constexpr int H = ((2 * TS + V - 1) / V) * V;   // TS invalid columns per side and march (TS - 1 cross-lane sweeps + the first sweep's DPP neighbours)
  const int j0 = c0 + lane * V;
  const int nx = g.nx, ny = g.ny;
  const int jlo = c0 + H > 1 ? c0 + H : 1;
  const int jhi = c0 + W - H - 1 < ny ? c0 + W - H - 1 : ny;
  const int64_t pitch = g.pitch;
  T an[V], as_[V], apI[V], yI[V];
#pragma unroll
  for (int q = 0; q < V; ++q) {
    const int j = j0 + q;
    an[q] = j != ny ? c.dyi2 : (T)0.0;
    as_[q] = j != 1 ? c.dyi2 : (T)0.0;
    apI[q] = (T)-1.0 * (c.dxi2 + c.dxi2 + an[q] + as_[q]);  // ap of rows 1 < i < nx
    yI[q] = (T)1 / apI[q];
    if (j < 1 || j > ny) yI[q] = (T)0;  // out-of-domain columns: every sweep yields the value 0
  }
  auto rowptr = [&](const T* base, int r) {
    const int rc = r < g.row_lo ? g.row_lo : (r > g.row_hi ? g.row_hi : r);
    return base + (size_t)(rc - g.row_lo) * (size_t)pitch + (size_t)(g.col0 + j0);
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
    for (int k = 0; k < 2 * TS + 2; ++k) __syncthreads();   // the steps the first march is ahead
    jp_get<T, V>(ring[0][0], lds.p[(t0 - 2) & 7], lane);
    jp_get<T, V>(ring[0][1], lds.p[(t0 - 1) & 7], lane);
    jp_get<T, V>(ring[0][2], lds.p[t0 & 7], lane);
    jp_get<T, V>(rq[0], lds.b[(t0 - 1) & 7], lane);
  }
  const T* const pn_tile = pn + (int64_t)(g.col0 + c0);
  const int voff_st = (j0 >= jlo && j0 + V - 1 <= jhi) ? lane * (int)(V * sizeof(T)) : kBufSkip;
  auto sub = [&](auto uc, int t) {
    constexpr int U = decltype(uc)::value;
    constexpr int kM = U % 3, kC = (U + 1) % 3, kE = (U + 2) % 3;
    T carry[V];
    if constexpr (ROLE == 0) jp_put<T, V>(lds.b[(t - 1) & 7], lane, rq[U % 6]);   // rhs row t - 1 (loaded an iteration ago) for the second march
#pragma unroll
    for (int s = 1; s <= TS; ++s) {
      const int i = t - s;
      const T sl = lane_up_z(ring[s - 1][kC][V - 1]);
      const T sr = lane_dn_z(ring[s - 1][kC][0]);
      if (s > 1) {
        const T coef = (i + 1 >= 1 && i + 1 <= nx) ? c.dxi2 : (T)0.0;
#pragma unroll
        for (int q = 0; q < V; ++q) ring[s - 1][kE][q] = coef * carry[q];
      }
      if (s > 1 && t < ra - TS + 2 * s) continue;
      const bool edge = (i == 1) || (i == nx);
      const T ae = i != nx ? c.dxi2 : (T)0.0;
      const T aw = i != 1 ? c.dxi2 : (T)0.0;
      const int slot = ((U + 1 - s) % 6 + 6) % 6;
      T num[V];
#pragma unroll
      for (int q = 0; q < V; ++q) {
        const T N = q == V - 1 ? sr : ring[s - 1][kC][q + 1];
        const T S = q == 0 ? sl : ring[s - 1][kC][q - 1];
        if (s > 1)
          num[q] = rq[slot][q] - ring[s - 1][kE][q] - ring[s - 1][kM][q] - N - S;
        else
          num[q] = rq[slot][q] - ae * ring[s - 1][kE][q] - aw * ring[s - 1][kM][q] - an[q] * N - as_[q] * S;
      }
      if (edge) {
#pragma unroll
        for (int q = 0; q < V; ++q) {
          const T ap = (T)-1.0 * (ae + aw + an[q] + as_[q]);
          T o = div_by_const<T>(num[q], ap, (T)1.0 / ap);
          if ((j0 + q) < 1 || (j0 + q) > ny) o = (T)0;
          carry[q] = o;
        }
      } else {
        div_by_const_v<T, V>(carry, num, apI, yI, &hit);
      }
      if (s == 1 && t < t1) {
        if constexpr (ROLE == 0) {
          load_c<T, V>(ring[0][kM], rowptr(p, t + 1));
          load_s<T, V>(rq[(U + 1) % 6], rowptr(rhs, t));
        } else {
          jp_get<T, V>(ring[0][kM], lds.p[(t + 1) & 7], lane);
          jp_get<T, V>(rq[(U + 1) % 6], lds.b[t & 7], lane);
        }
      }
    }
    const int io = t - TS;
    if constexpr (ROLE == 0) {
      if (io >= ra && io <= rb) jp_put<T, V>(lds.p[io & 7], lane, carry);
    } else if constexpr (BS) {
      store_buf_nt<T, V>(pn_tile, (io >= ra && io <= rb) ? voff_st : kBufSkip,
                         (io >= ra && io <= rb) ? (int)((int64_t)(io - g.row_lo) * pitch * (int64_t)sizeof(T)) : 0, carry);
    } else if (io >= ra && io <= rb) {
      store_s<T, V>(pn + at(g, io, j0), carry, j0, jlo, jhi);
    }
    __syncthreads();
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
}

template <typename T, int V, int TS, bool BS>
__global__ __launch_bounds__(128) void k_jacobi_pair(Geom g, Consts<T> c, const T* __restrict__ p,
                                                     const T* __restrict__ rhs, T* __restrict__ pn, int R, int ntt,
                                                     TbPlan tp, int first, int last) {
  constexpr int W = 64 * V;
  constexpr int H = ((2 * TS + V - 1) / V) * V;   // TS invalid columns per side and march (TS - 1 cross-lane sweeps + the first sweep's DPP neighbours)
  constexpr int STRIDE = W - 2 * H;
  __shared__ __attribute__((aligned(16))) JpRing<T, V> lds;
  if (last < first) { first = g.ilo; last = g.ihi; }
  const int pair = (int)blockIdx.x;
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  // pair -> (tile column tj, rows [ra, rb]): chunks of R rows of every column, or the equal-cost chunks of the step's plan
  int tj = pair % ntt, ra = first + (pair / ntt) * R, rb = ra + R - 1;
  bool planned = false;
  if (tp.masks != nullptr) {
    const unsigned long long* pl = tp.plan;
    if (pl[0] != 0ull) {
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
  int hit = 0;
  if (role == 0) {
    jacobi_pair_march<T, V, TS, BS, 0>(g, c, p, rhs, pn, lds, c0, lane, ra - TS, rb + TS, hit);
  } else {
    jacobi_pair_march<T, V, TS, BS, 1>(g, c, p, rhs, pn, lds, c0, lane, ra, rb, hit);
  }
  if (tp.masks != nullptr && __any(hit != 0) && lane == 0) {   // report the (row band, tile column) cells of this chunk
    for (int b = tb_band_of(g, ra < g.ilo ? g.ilo : ra); b <= tb_band_of(g, rb > g.ihi ? g.ihi : rb); ++b)
      atomicOr(tp.masks + tb_word(tp.par ^ 1, b, tj), 1ull << (tj & 63));
  }
}

}  // namespace vof
