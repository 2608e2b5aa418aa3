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
// instead of memory) and the chunk bounds of the pair.  The arithmetic of a row is ONE source shared with the stand-alone
// kernels (round 6): MomentumWindow::step (kernels/momentum.h), TransportRow (kernels/transport.h).
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
  // the register window and the arithmetic of an iteration are k_transport's (TransportWindow, kernels/transport.h): what differs
  // here is where u, v and F'' go -- the pair's ring in LDS (and memory where somebody reads them) -- and the chunk bounds of the pair
  TransportWindow<T, V> win;
  {
    T f1[V], pr1[V], v0[V];
    load_c<T, V>(f1, rowptr(F, tra - 3));
    load_c<T, V>(pr1, rowptr(p, tra - 3));
    const bool swept = YFIRST && (IN || tra - 3 >= ilo);
    if (swept) {
      load_s<T, V>(v0, rowptr(vs, tra - 3));
    } else {
#pragma unroll
      for (int q = 0; q < V; ++q) v0[q] = (T)0;
    }
    win.template init<YFIRST, IN>(c, j0, ny, f1, pr1, v0, swept);
  }
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
      T out[V];
      const bool own = r >= ma && r <= mb;
      win.template step<YFIRST, IN>(c, r, ilo, ihi, nx, j0, ny, Fr, ur, vr, pr, out, tra, trb, [&](bool urow) {
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
      });
      const int io = r - 3;
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
  bool dom[V];
#pragma unroll
  for (int q = 0; q < V; ++q) dom[q] = IN || ((j0 + q) >= 1 && (j0 + q) <= ny);
  auto mirrow = [&](int r) { return IN ? r : (r == 0 ? 1 : (r == nx + 1 ? nx : r)); };   // virtual ghost rows of F and v (:176-189)
  const bool edge_cols = !IN && (c0 - 1 <= 0 || c0 + W >= ny + 1);
  auto get_F = [&](Row<T, V>& w, int r) { ring_get<T, V>(w, ring.f[mirrow(r) & 7], lane); };
  auto get_u = [&](Row<T, V>& w, int r) { ring_get<T, V>(w, ring.u[r & 7], lane); };
  auto get_v = [&](Row<T, V>& w, int r) { ring_get<T, V>(w, ring.v[mirrow(r) & 7], lane); };
  // the register window and the arithmetic of an iteration are k_momentum's (MomentumWindow, kernels/momentum.h): what differs
  // here is where the rows come from -- the pair's ring in LDS -- and the chunk bounds of the pair
  MomentumWindow<T, V> win;
  win.init(c);
  const int r0 = ma - 1, r1 = mb + 3;
  const T* const us_tile = us_out + (int64_t)(g.col0 + c0);
  const T* const vs_tile = vs_out + (int64_t)(g.col0 + c0);
  const T* const rhs_tile = rhs + (int64_t)(g.col0 + c0);
  const int voff_st = (j0 >= jlo && j0 + V - 1 <= jhi) ? lane * (int)(V * sizeof(T)) : kBufSkip;
  Row<T, V> un, vn;   // u / v row r-1 of the coming iteration
  MomentumWindow<T, V>::zero_row(un); MomentumWindow<T, V>::zero_row(vn);
  for (int t = t_lo; t <= t_hi; ++t) {
    const int r = t - 5;
    if (r == r0) {   // the window's first two rows (k_momentum loads them in front of its loop)
      Row<T, V> f2, f1;
      get_F(f2, r0 - 2);
      get_F(f1, r0 - 1);
      if (edge_cols) {
        mirror_ghost_cols<T, V>(f2, j0, ny);
        mirror_ghost_cols<T, V>(f1, j0, ny);
      }
      win.set_F(f2, f1);
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
      const int i = r - 2, i3 = r - 3;
      T out[V];
      win.template step<IN, ABL>(c, r, ilo, ihi, j0, ny, dom, F0, u1, v1, out, BS || (i3 >= ma && i3 <= mb), [&](const T (&us2)[V], const T (&vs2)[V]) {
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
      });
      if constexpr ((ABL & ABL_NO_STORE) != 0) {
        if (BS || (i3 >= ma && i3 <= mb)) asm volatile("" :: "v"(out[0]), "v"(out[V - 1]));
      } else if constexpr (BS) {
        store_buf_nt<T, V>(rhs_tile, (i3 >= ma && i3 <= mb) ? voff_st : kBufSkip,
                           (i3 >= ma && i3 <= mb) ? (int)((int64_t)(i3 - g.row_lo) * g.pitch * (int64_t)sizeof(T)) : 0, out);
      } else if (i3 >= ma && i3 <= mb) {
        store_s<T, V>(rhs + at(g, i3, j0), out, j0, jlo, jhi);
      }
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
