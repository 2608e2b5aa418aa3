// kernels/plan.h -- the equal-cost work plan of k_jacobi_tb (tb_make_plan)
//
// Part of the gfx950 kernel set of the 2-D VOF hot path (see vof2d_kernels.h for the conventions:
// reference line citations, expression order, one wave = 64*V columns marching along i).
#pragma once
#include "common.h"

namespace vof {

// ------------------------------------------------------------------ work plan of k_jacobi_tb
// While the decaying front of the pressure iteration crosses the grid, the waves of k_jacobi_tb
// whose rows lie in the band of tiny values (1e-280 ... 4.9e-324) execute about twice the
// instructions per row (the exact division's scaled tier), and with one residency round per launch
// they run on alone after the others have ended: 145 us per launch instead of 92 (4096^2).  The
// launches therefore report WHERE the tier ran -- one bit per (row band, tile column) -- and the next
// step cuts every tile column into chunks of equal COST instead of equal length: the same number of
// waves, short chunks inside the band, slightly longer ones elsewhere, so that all waves end
// together again.  Which rows a wave takes never changes a value (every cell is computed from the
// same operands whatever the chunking; the parity tests run with the plan active).
//   TbPlan::masks  two sets of TB_BANDS x (TB_COLS / 64) words, one bit per tile column: a step reads set (istep & 1)
//                  -- what the previous step's launches reported -- and reports into the other, which
//                  this step's planner clears first (its last readers were the previous step's launches)
//   TbPlan::plan   [0] = 1 if a plan is active (else the uniform layout), [1 + wave] = the wave's
//                  tile column and rows, packed (plan_pack)
// The planner is one extra block -- the first -- of k_momentum's launch (the kernel in front of the
// Jacobi launches in the fused step): it runs beside the other blocks, off the critical path.
constexpr int TB_BANDS = 64;          // row bands of the hit masks
constexpr int TB_COLS = 128;          // tile columns the masks cover (two 64-bit words per band): grids up to ~14 800 wide
constexpr int TB_SLOW10 = 20;         // cost of a band row in tenths of an ordinary row (k_jacobi_tb; the pair kernel's: TbPlan::slow10)
struct TbPlan {
  unsigned long long* masks;          // nullptr: no plan (uniform layout)
  unsigned long long* plan;
  int ntt, R, waves, par;             // tile columns (<= TB_COLS), uniform chunk length, waves of a launch, istep & 1
  int slow10;                         // cost of a row of a reported band in tenths of an ordinary row (0: TB_SLOW10)
};
// plan[0] of an active plan: 1 + the geometry it was planned for.  A launch reads the plan only if that is its own
// geometry (a plan of k_jacobi_tb's tile columns read by k_jacobi_pair would leave rows out); anything else is "no plan".
__device__ __forceinline__ unsigned long long plan_key(const TbPlan& tp) {
  return 1ull | ((unsigned long long)(unsigned)tp.waves << 1) | ((unsigned long long)(unsigned)tp.R << 33) | ((unsigned long long)(unsigned)tp.ntt << 48);
}
__device__ __forceinline__ unsigned long long plan_pack(int tj, int ra, int rb) {
  return (unsigned long long)(unsigned)tj | ((unsigned long long)(unsigned)ra << 8) | ((unsigned long long)(unsigned)rb << 36);
}
__device__ __forceinline__ int tb_band_of(const Geom& g, int i) {   // row -> band index
  const int rows = g.ihi - g.ilo + 1, h = (rows + TB_BANDS - 1) / TB_BANDS;
  return (i - g.ilo) / h;
}
// word index of (mask set, band, tile column) and the column's bit in it
__device__ __forceinline__ int tb_word(int set, int b, int tj) { return (set * TB_BANDS + b) * (TB_COLS / 64) + (tj >> 6); }
// One block of 256 threads (the planner block of k_momentum's launch; it must not outlast the
// launch's other waves, so the per-chunk work is spread over all its threads).  32-bit integers.
// 3 KB of LDS: every block of the launch reserves it, so it is kept small (a per-column cost prefix
// table, 33 KB, capped k_momentum at 4 blocks per CU; the prefix is now a closed form of the column's
// 64 band bits, tb_prefix).
struct TbPlanShared {
  unsigned long long band[TB_BANDS][TB_COLS / 64];   // the reported (band, column) bits, as the launches wrote them
  unsigned long long col[TB_COLS];             // the same bits per tile column: bit b = band b of column j was reported
  int first[TB_COLS + 1];                      // first wave of column j; first[TB_COLS] = planned waves
  int n[TB_COLS];                              // chunks of column j
};
__device__ __forceinline__ bool tb_bit(const TbPlanShared& sh, int b, int j) { return ((sh.band[b][j >> 6] >> (j & 63)) & 1ull) != 0ull; }
// cost of rows [0, min(b * bh, rows)) of a tile column, in tenths of a row: 10 per row, TB_SLOW10 per row of a
// reported band.  Bands 0 .. rows / bh - 1 are bh rows long, the next one holds the remainder, the rest are empty.
__device__ __forceinline__ unsigned tb_prefix(unsigned long long colbits, int b, int rows, int bh, int slow10) {
  const int nfull = rows / bh, part = rows - nfull * bh;       // (bh >= 1: rows >= 1 on every handle)
  const unsigned long long upto = b >= 64 ? ~0ull : ((1ull << b) - 1ull);
  const unsigned long long full = nfull >= 64 ? ~0ull : ((1ull << nfull) - 1ull);
  int slow_rows = bh * __popcll(colbits & upto & full);
  if (part > 0 && nfull < b && nfull < 64 && ((colbits >> nfull) & 1ull)) slow_rows += part;
  const int before = b * bh < rows ? b * bh : rows;
  return 10u * (unsigned)before + (unsigned)(slow10 - 10) * (unsigned)slow_rows;
}
// row position (0 .. rows) where the cumulative cost of column j reaches T
__device__ __forceinline__ int tb_pos(const TbPlanShared& sh, int j, unsigned T, int rows, int bh, int slow10) {
  const unsigned long long colbits = sh.col[j];
  int lo = 0, hi = TB_BANDS;           // largest b with prefix(b) <= T
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tb_prefix(colbits, mid, rows, bh, slow10) <= T) lo = mid; else hi = mid;
  }
  const bool slow = ((colbits >> lo) & 1ull) != 0ull;
  const unsigned rest = T - tb_prefix(colbits, lo, rows, bh, slow10);
  int pos = lo * bh + (int)(slow ? rest / (unsigned)slow10 : rest / 10u);
  const int bend = (lo + 1) * bh;
  if (pos > bend) pos = bend;
  return pos < rows ? pos : rows;
}
__device__ __forceinline__ int tb_wave_sum(int v) {
  for (int sft = 32; sft > 0; sft >>= 1) v += __shfl_xor(v, sft, 64);
  return v;
}
__device__ void tb_make_plan(const Geom& g, const TbPlan& tp, TbPlanShared& sh) {
  constexpr int CW = TB_COLS / 64;
  const int slow10 = tp.slow10 > 10 ? tp.slow10 : TB_SLOW10;
  const int t = threadIdx.x, lane = t & 63;
  const int rows = g.ihi - g.ilo + 1, bh = (rows + TB_BANDS - 1) / TB_BANDS;
  unsigned long long mine[CW];                  // band `lane` (nobody writes the read set during this step)
  bool some = false;
#pragma unroll
  for (int w = 0; w < CW; ++w) {
    mine[w] = tp.masks[tb_word(tp.par, lane, 0) + w];
    some = some || mine[w] != 0ull;
  }
  const bool any = __any(some);                 // (the same in all four waves)
  if (t < 64) {
#pragma unroll
    for (int w = 0; w < CW; ++w) {
      tp.masks[tb_word(tp.par ^ 1, lane, 0) + w] = 0ull;   // this step's launches report into the other set
      sh.band[lane][w] = mine[w];
    }
    if (lane == 0) tp.plan[0] = any ? plan_key(tp) : 0ull;
  }
  if (!any) return;                             // block-uniform
  __syncthreads();
  if (t < 64) {   // wave 0: per column (lane j and j + 64), the cost prefix over the bands and the number of chunks
    unsigned cost[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      const int j = lane + 64 * c;
      unsigned long long bits = 0ull;
      for (int b = 0; b < TB_BANDS; ++b) bits |= tb_bit(sh, b, j) ? (1ull << b) : 0ull;
      sh.col[j] = bits;
      cost[c] = j < tp.ntt ? tb_prefix(bits, TB_BANDS, rows, bh, slow10) : 0u;
    }
    unsigned total = 0;
#pragma unroll
    for (int c = 0; c < CW; ++c) total += (unsigned)tb_wave_sum((int)cost[c]);
    // chunks per column, proportional to its cost (at least one), within the waves of a launch
    const int nmax = rows >= 4 ? rows / 4 : 1;
    int n[CW], sum = 0;
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      const int j = lane + 64 * c;
      n[c] = j < tp.ntt ? (int)(((unsigned long long)cost[c] * (unsigned)tp.waves) / total) : 0;
      if (j < tp.ntt && n[c] < 1) n[c] = 1;
      if (n[c] > nmax) n[c] = nmax;
      sum += tb_wave_sum(n[c]);
    }
    // (the floor leaves a few waves over: one more for the first columns; never more than `waves`)
    const int left = tp.waves - sum;
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      const int j = lane + 64 * c;
      if (left > 0 && j < tp.ntt && j < left && n[c] < nmax) n[c] += 1;
    }
    for (int guard = 0; guard < 8192; ++guard) {   // the at-least-one rule can overshoot on tiny grids: trim the largest
      sum = 0;
#pragma unroll
      for (int c = 0; c < CW; ++c) sum += tb_wave_sum(n[c]);
      if (sum <= tp.waves) break;
      int mx = 0;
#pragma unroll
      for (int c = 0; c < CW; ++c) mx = n[c] > mx ? n[c] : mx;
      for (int sft = 32; sft > 0; sft >>= 1) { const int o = __shfl_xor(mx, sft, 64); mx = o > mx ? o : mx; }
      bool done = false;                         // the first column holding the maximum gives one up
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        const unsigned long long who = __ballot(!done && n[c] == mx);
        if (who != 0ull) {
          if (!done && lane == __ffsll((long long)who) - 1) n[c] -= 1;
          done = true;
        }
      }
    }
    int base = 0;   // prefix sums over the 64-column halves -> every column's first wave
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      const int j = lane + 64 * c;
      int incl = n[c];
      for (int sft = 1; sft < 64; sft <<= 1) { const int o = __shfl_up(incl, sft, 64); if (lane >= sft) incl += o; }
      sh.n[j] = n[c];
      sh.first[j] = base + incl - n[c];
      base += __shfl(incl, 63, 64);
    }
    if (lane == 0) sh.first[TB_COLS] = base;
  }
  __syncthreads();
  const int planned = sh.first[TB_COLS];
  for (int w = t; w < tp.waves; w += (int)blockDim.x) {
    unsigned long long e = plan_pack(0, 1, 0);   // waves past the planned ones: empty
    if (w < planned) {
      int lo = 0, hi = TB_COLS;                  // the column of wave w: largest j with first[j] <= w
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (sh.first[mid] <= w) lo = mid; else hi = mid;
      }
      const int j = lo, n = sh.n[j], k = w - sh.first[j];
      // chunk k of column j: between the rows where the cumulative cost reaches k / n and (k + 1) / n of the column's
      const unsigned cost = tb_prefix(sh.col[j], TB_BANDS, rows, bh, slow10);
      const int a = k == 0 ? 0 : tb_pos(sh, j, (unsigned)(((unsigned long long)cost * (unsigned)k) / (unsigned)n), rows, bh, slow10);
      const int b = k == n - 1 ? rows : tb_pos(sh, j, (unsigned)(((unsigned long long)cost * (unsigned)(k + 1)) / (unsigned)n), rows, bh, slow10);
      e = plan_pack(j, g.ilo + a, g.ilo + b - 1);   // (b == a: an empty chunk, the wave returns at once)
    }
    tp.plan[1 + w] = e;
  }
}

}  // namespace vof
