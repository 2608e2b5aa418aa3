// kernels/common.h -- tile rows, streaming loads / stores, DPP neighbours, XCD block order, exact division by a lane constant
//
// Part of the gfx950 kernel set of the 2-D VOF hot path (see vof2d_kernels.h for the conventions:
// reference line citations, expression order, one wave = 64*V columns marching along i).
#pragma once
#include "../vof2d_device.h"

namespace vof {

// ------------------------------------------------------------------ helpers
// Diagnostic build only (-DVOF_WAVE_TIMES, tools/wave_balance.py): every wave of the selected kernel
// records when it started and ended (s_memrealtime, 100 MHz), which shows how evenly a launch's
// waves finish.  The product build compiles WaveTimer to nothing.
#ifdef VOF_WAVE_TIMES
__device__ unsigned long long* vof_wave_times = nullptr;  // [2 * wave] = start, [2 * wave + 1] = end; behind them (from 2 * cap): cycles inside barriers, cycles in all
__device__ int vof_wave_kid = -1;
__device__ unsigned int vof_wave_cap = 0;
struct WaveTimer {
  unsigned long long t0, c0, wait;
  unsigned int wave;
  bool on;
  __device__ __forceinline__ WaveTimer(int kid) {
    on = vof_wave_times != nullptr && vof_wave_kid == kid;
    wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    t0 = on ? wall_clock64() : 0ull;
    c0 = on ? __builtin_readcyclecounter() : 0ull;
    wait = 0ull;
  }
  // the workgroup barrier of the pair kernels, with the cycles this wave spent inside it
  __device__ __forceinline__ void barrier() {
    if (on) {
      const unsigned long long t = __builtin_readcyclecounter();
      __syncthreads();
      wait += __builtin_readcyclecounter() - t;
    } else {
      __syncthreads();
    }
  }
  __device__ __forceinline__ ~WaveTimer() {
    if (on && (threadIdx.x & 63) == 0 && wave < vof_wave_cap) {
      vof_wave_times[2 * wave] = t0;
      vof_wave_times[2 * wave + 1] = wall_clock64();
      vof_wave_times[2 * vof_wave_cap + 2 * wave] = wait;
      vof_wave_times[2 * vof_wave_cap + 2 * wave + 1] = __builtin_readcyclecounter() - c0;
    }
  }
};
#else
struct WaveTimer {
  __device__ __forceinline__ WaveTimer(int) {}
  __device__ __forceinline__ void barrier() { __syncthreads(); }
};
#endif
// Diagnostic build only (-DVOF_SHORTCUT_STATS, tools/probes/shortcut_stats.py): how often the wave-level shortcuts of
// k_momentum and k_transport are taken (one count per wave and row).  The product build compiles VOF_STAT to nothing.
#ifdef VOF_SHORTCUT_STATS
__device__ unsigned long long vof_stats[16];
#define VOF_STAT(k) do { if ((threadIdx.x & 63) == 0) atomicAdd(&vof::vof_stats[k], 1ull); } while (0)
#else
#define VOF_STAT(k) do { } while (0)
#endif
enum : int { WT_MOMENTUM = 0, WT_JACOBI_TB = 3, WT_FCT_X = 5, WT_FCT_Y = 6, WT_JACOBI = 2, WT_TRANSPORT = 12, WT_JACOBI_PAIR = 13, WT_TM = 14 };  // = KernelId of the runtime
// Diagnostic build only (tools/probes/pair_bound.py, timing with WRONG values): what a pair kernel's time hangs on.
// Bits of the ABL template argument of k_jacobi_pair / k_tm; the product instantiates ABL = 0 only.
enum : int {
  ABL_FIXED_ROW = 1,   // the first wave loads the same row over and over (L1 / L2 hits instead of the HBM stream)
  ABL_NO_STORE = 2,    // no global store
  ABL_PRIO0 = 4,       // NO s_setprio 1 for the first wave of the pair (the product runs it at priority 1)
  ABL_PRIO1 = 8,       // s_setprio 1 for the second
  ABL_IDLE1 = 16,      // the second wave only keeps the barriers
  ABL_PASS0 = 32,      // the first wave hands its input rows on without computing
  ABL_NO_UPRED = 64,   // k_tm: the momentum wave leaves out the advection / diffusion terms of u* (what would another wave's taking them over buy?)
  ABL_NO_VPRED = 128   // ... of v* too
};
template <typename T, int V>
struct Row {  // one row of a wave tile as seen by a lane: j0-1 | j0..j0+V-1 | j0+V
  T l;
  T c[V];
  T r;
};

template <typename T, int V>
__device__ __forceinline__ void load_c(T (&c)[V], const T* __restrict__ p) {
  Pack<T, V> k = *reinterpret_cast<const Pack<T, V>*>(p);
#pragma unroll
  for (int q = 0; q < V; ++q) c[q] = k.v[q];
}
template <typename T, int V>
__device__ __forceinline__ void load_row(Row<T, V>& w, const T* __restrict__ p) {
  load_c<T, V>(w.c, p);
  w.l = p[-1];
  w.r = p[V];
}
// streaming (nontemporal) forms for data touched once per launch
template <typename T, int V>
__device__ __forceinline__ void load_c_nt(T (&c)[V], const T* __restrict__ p) {
  typedef T vec_t __attribute__((ext_vector_type(V)));
  vec_t k = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(p));
#pragma unroll
  for (int q = 0; q < V; ++q) c[q] = k[q];
}
template <typename T, int V>
__device__ __forceinline__ void store_c_nt(T* __restrict__ p, const T (&c)[V], int j0, int jlo, int jhi) {
  typedef T vec_t __attribute__((ext_vector_type(V)));
  if (j0 >= jlo && j0 + V - 1 <= jhi) {
    vec_t k;
#pragma unroll
    for (int q = 0; q < V; ++q) k[q] = c[q];
    __builtin_nontemporal_store(k, reinterpret_cast<vec_t*>(p));
  } else {
#pragma unroll
    for (int q = 0; q < V; ++q)
      if (j0 + q >= jlo && j0 + q <= jhi) p[q] = c[q];
  }
}
// store columns j0..j0+V-1 restricted to [jlo, jhi]
template <typename T, int V>
__device__ __forceinline__ void store_c(T* __restrict__ p, const T (&c)[V], int j0, int jlo, int jhi) {
  if (j0 >= jlo && j0 + V - 1 <= jhi) {
    Pack<T, V> k;
#pragma unroll
    for (int q = 0; q < V; ++q) k.v[q] = c[q];
    *reinterpret_cast<Pack<T, V>*>(p) = k;
  } else {
#pragma unroll
    for (int q = 0; q < V; ++q)
      if (j0 + q >= jlo && j0 + q <= jhi) p[q] = c[q];
  }
}
// Streaming forms: arrays that a launch reads or writes exactly once (kernel outputs, rhs, u*, v*)
// carry the nontemporal hint, so they do not displace the row halos that vertically adjacent chunks
// share through L2 (k_jacobi at 4096^2 fp64: 79.5 -> 73.0 us).
template <typename T, int V>
__device__ __forceinline__ void load_s(T (&c)[V], const T* __restrict__ p) { load_c_nt<T, V>(c, p); }
template <typename T, int V>
__device__ __forceinline__ void store_s(T* __restrict__ p, const T (&c)[V], int j0, int jlo, int jhi) {
  store_c_nt<T, V>(p, c, j0, jlo, jhi);
}
// Lane-masked streaming store WITHOUT a branch: a raw buffer store whose lanes carry a byte offset into a 2 GiB
// window behind `base` (wave-uniform); a lane whose offset is kBufSkip lies outside the window and the hardware
// drops its write.  Unlike an exec-masked global store (which the compiler wraps in an s_cbranch_execz it then
// cannot count) this is one unconditional memory instruction, so the s_waitcnt counts of a software-pipelined loop
// stay exact: waiting for the rows requested an iteration ago does not also wait for the stores issued since.
constexpr int kBufSkip = (int)0x80000000;
// (the descriptor and the row offset live in SGPRs: values the compiler cannot prove wave-uniform would be read back
// lane by lane in a waterfall loop around every store, so they are declared uniform here)
template <typename T>
__device__ __forceinline__ T* wave_uniform(T* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (T*)(((unsigned long long)hi << 32) | lo);
}
template <typename T, int V>
__device__ __forceinline__ void store_buf_nt(const T* base, int lane_off_bytes, int row_off_bytes, const T (&c)[V]) {
  static_assert(sizeof(T) * V == 16 || sizeof(T) * V == 8, "one b128 / b64 store per lane");
  // The row offset travels in the lane offsets (kBufSkip + a row offset still lies outside the window), not in the
  // instruction's scalar offset: with an SGPR soffset the compiler assumes that the VGPRs holding more than 8 bytes
  // of store data may be overwritten by the very next VALU instruction, and on gfx950 they may not -- k_tm's rhs
  // store picked up, in some lanes and not every time, the v* values a v_mov wrote into its data registers one
  // instruction later.  With soffset = 0 the hazard recognizer inserts the wait state itself.
  row_off_bytes = __builtin_amdgcn_readfirstlane(row_off_bytes);
  lane_off_bytes += row_off_bytes;
  asm volatile("" : "+v"(lane_off_bytes));   // (or the instruction selection splits the uniform part off into soffset again)
  row_off_bytes = 0;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(wave_uniform(const_cast<T*>(base)), 0, kBufSkip, 0x00020000);
  typedef T vec_t __attribute__((ext_vector_type(V)));
  vec_t k;
#pragma unroll
  for (int q = 0; q < V; ++q) k[q] = c[q];
  if constexpr (sizeof(T) * V == 16) {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, k), r, lane_off_bytes, row_off_bytes, 2 /* nt */);
  } else {
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, k), r, lane_off_bytes, row_off_bytes, 2 /* nt */);
  }
}
// The load that goes with it: the lanes' byte offsets (wave-constant from row to row) in a VGPR, the row's byte offset in
// an SGPR, the base in the descriptor -- no 64-bit address arithmetic per row and lane (a field below 2 GiB, as above).
template <typename T, int V>
__device__ __forceinline__ void load_buf(T (&c)[V], const T* base, int lane_off_bytes, int row_off_bytes, bool nt) {
  static_assert(sizeof(T) * V == 16 || sizeof(T) * V == 8, "one b128 / b64 load per lane");
  row_off_bytes = __builtin_amdgcn_readfirstlane(row_off_bytes);
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(wave_uniform(const_cast<T*>(base)), 0, kBufSkip, 0x00020000);
  typedef T vec_t __attribute__((ext_vector_type(V)));
  vec_t k;
  if constexpr (sizeof(T) * V == 16) {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const u4 w = nt ? __builtin_amdgcn_raw_buffer_load_b128(r, lane_off_bytes, row_off_bytes, 2) : __builtin_amdgcn_raw_buffer_load_b128(r, lane_off_bytes, row_off_bytes, 0);
    k = __builtin_bit_cast(vec_t, w);
  } else {
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    const u2 w = nt ? __builtin_amdgcn_raw_buffer_load_b64(r, lane_off_bytes, row_off_bytes, 2) : __builtin_amdgcn_raw_buffer_load_b64(r, lane_off_bytes, row_off_bytes, 0);
    k = __builtin_bit_cast(vec_t, w);
  }
#pragma unroll
  for (int q = 0; q < V; ++q) c[q] = k[q];
}
template <typename T, int V>
__device__ __forceinline__ T left_of(const Row<T, V>& w, int q) { return q == 0 ? w.l : w.c[q - 1]; }
template <typename T, int V>
__device__ __forceinline__ T right_of(const Row<T, V>& w, int q) { return q == V - 1 ? w.r : w.c[q + 1]; }

// XCD-contiguous block order: physical workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md "Workgroup
// dispatch"; observed, not promised -- only speed depends on it) and each XCD has its own 4 MiB L2.
// Mapped through this, XCD x works through ONE contiguous range of logical blocks, so tiles that
// share cache lines (column overlap, halo rows) mostly share an L2.  Used by k_jacobi_tb only, whose
// launch is one residency round: 55.5 -> 53.9 us per launch on a 1056 x 8192 strip (59.3 -> 55.4 inside
// the tiny-value front), neutral at 4096^2 and 2048^2.  The multi-round kernels must NOT use it: their
// blocks are dispatched in index order, and eight separate bands in flight instead of one compact band
// cost k_momentum 17 % and k_transport 32 % (profiles/r03_ab_xcd_and_strip_chunks.log).
__device__ __forceinline__ int xcd_contiguous_block(int b, int n) {
  const int q = n >> 3, r = n & 7, x = b & 7;
  return x * q + (x < r ? x : r) + (b >> 3);
}
// The pair kernels' order: windows of 8 * kXcdGroup consecutive blocks (in flight together, in index order: one compact
// band), inside a window XCD x takes kXcdGroup consecutive logical blocks -- horizontally adjacent tiles, which share
// their overlap columns through that XCD's L2.  The last, partial window keeps its order.
constexpr int kXcdGroup = 10;
__device__ __forceinline__ int xcd_grouped_block(int b, int n) {
  constexpr int WIN = 8 * kXcdGroup;
  const int base = (b / WIN) * WIN;
  if (base + WIN > n) return b;
  const int o = b - base;
  return base + (o & 7) * kXcdGroup + (o >> 3);
}

// wave -> (column tile, row chunk).  Rows [first, last] are split in chunks of R.
template <int V>
__device__ __forceinline__ bool wave_tile(const Geom& g, int first, int last, int R, int& j0, int& ra,
                                          int& rb) {
  const int wave = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // SGPR: rows are wave-uniform
  const int lane = threadIdx.x & 63;
  const int tj = wave % g.ntj;
  const int ch = wave / g.ntj;
  j0 = 1 + tj * 64 * V + lane * V;
  ra = first + ch * R;
  rb = ra + R - 1 < last ? ra + R - 1 : last;
  return ra <= last && j0 <= g.ny;
}
__device__ __forceinline__ size_t at(const Geom& g, int i, int j) {
  return (size_t)(i - g.row_lo) * (size_t)g.pitch + (size_t)(g.col0 + j);
}

// ------------------------------------------------------------------ cross-lane neighbours (DPP)
// lane_up(x): value of lane-1 (lane 0 keeps its own); lane_dn(x): value of lane+1 (lane 63 keeps
// its own).  gfx9 DPP wave_shr:1 / wave_shl:1 -- a VALU move, no LDS round trip like ds_bpermute.
__device__ __forceinline__ int dpp_up(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int dpp_dn(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ double lane_up(double x) {
  return __hiloint2double(dpp_up(__double2hiint(x)), dpp_up(__double2loint(x)));
}
__device__ __forceinline__ double lane_dn(double x) {
  return __hiloint2double(dpp_dn(__double2hiint(x)), dpp_dn(__double2loint(x)));
}
__device__ __forceinline__ float lane_up(float x) { return __int_as_float(dpp_up(__float_as_int(x))); }
__device__ __forceinline__ float lane_dn(float x) { return __int_as_float(dpp_dn(__float_as_int(x))); }
// Zero-filling forms (bound_ctrl): lane 0 / lane 63 receive 0 instead of keeping their own value,
// which lets the move read its source register directly (no copy first).  For kernels whose tile
// edge columns are recomputed by the neighbouring tile anyway.
__device__ __forceinline__ int dpp_up_z(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ int dpp_dn_z(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true); }
__device__ __forceinline__ double lane_up_z(double x) {
  return __hiloint2double(dpp_up_z(__double2hiint(x)), dpp_up_z(__double2loint(x)));
}
__device__ __forceinline__ double lane_dn_z(double x) {
  return __hiloint2double(dpp_dn_z(__double2hiint(x)), dpp_dn_z(__double2loint(x)));
}
__device__ __forceinline__ float lane_up_z(float x) { return __int_as_float(dpp_up_z(__float_as_int(x))); }
__device__ __forceinline__ float lane_dn_z(float x) { return __int_as_float(dpp_dn_z(__float_as_int(x))); }

// ------------------------------------------------------------------ exact division by a lane constant
// a / b for a denominator that is constant per lane (ap of the Jacobi stencil).  With
// y = RN(1/b):  q = RN(a*y);  r = a - b*q (exact, one FMA);  RN(q + r*y) is the correctly rounded
// quotient (Markstein 1990; the same final step the hardware division expansion performs after
// its Newton iterations), i.e. bit-identical to IEEE a / b, for 3 FMA-rate ops instead of ~11.
// Outside a safe exponent window the remainder r would underflow (tiny a) or q overflow (huge a
// with |b| < 1).  There the numerator is scaled by an exact power of two, divided the same way and
// the quotient Q scaled back:
//   * tiny a (the decaying front of the Jacobi iteration walks through 1e-289 ... 4.9e-324 on its
//     way to exact zero): Q * 2^-k is exact while the quotient is normal.  A subnormal quotient is
//     rounded a second time by that multiplication; the two roundings differ from the single
//     IEEE one only if Q sits exactly on a midpoint of the subnormal grid (midpoints are
//     representable, and RN is monotonic) while the true quotient lies beside it -- the sign of
//     the exact remainder A - b*Q tells on which side, and the tie break is undone if it went the
//     other way;
//   * huge a: Q * 2^k is exact or overflows to the same infinity a / b rounds to.
// a == 0 gives the signed zero of a*y, an infinite a the infinity a*y, a NaN numerator NaN: every
// input gets the IEEE quotient without the ~11-op hardware expansion (and without a call, which
// would cost the register-heavy kernels their allocation).
template <typename T> struct DivLimits;
template <> struct DivLimits<double> {
  // lo: below it the remainder a - b*q of the fast form may fall off the subnormal grid (|a| < 2^-969) or the quotient be subnormal
  // (|a| < 2^-1022 |b|): 1e-289 = 2^-960 leaves ten binades for |b| up to 2^52.  (1e-280 until round 5: the band of the decaying
  // front that takes the scaled tier is a fifth narrower.)
  static constexpr double lo = 1e-289, hi = 1e280, up = 0x1p+256, dn = 0x1p-256, qmin = 0x1p-766 /* 2^-1022 * up */,
                          denorm_min = 0x1p-1074, half_step = 0x1p-819 /* denorm_min * up / 2 */,
                          inf = __builtin_huge_val();
};
template <> struct DivLimits<float> {
  static constexpr float lo = 1e-25f, hi = 1e25f, up = 0x1p+96f, dn = 0x1p-96f, qmin = 0x1p-30f /* 2^-126 * up */,
                         denorm_min = 0x1p-149f, half_step = 0x1p-54f, inf = __builtin_huge_valf();
};
template <typename T> __device__ __forceinline__ T dfma(T a, T b, T c);
template <> __device__ __forceinline__ double dfma<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <> __device__ __forceinline__ float dfma<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

template <typename T>
__device__ __forceinline__ T div_scaled(T a, T b, T y, T scale) {  // RN((a * scale) / b), scale = 2^+-k
  const T A = a * scale;
  const T Q0 = A * y;
  return dfma<T>(dfma<T>(-b, Q0, A), y, Q0);
}

// The tiny tier, |a| < lo, WITHOUT a branch (round 6): the scaled quotient, its rounding onto the subnormal grid and the repair of
// the double rounding are a straight run of a dozen instructions with three selects at the end -- as nested branches (rounds 1-5)
// the same arithmetic carried as many scalar instructions again (exec masks saved, tested, restored at three levels), in kernels
// whose waves issue one instruction at a time.  |Q| >= qmin needs no test: there Q * dn is exact, diff is 0 and nothing is repaired.
// An exact zero returns a * y (the signed zero of the quotient).
template <typename T>
__device__ __forceinline__ T div_tiny(T a, T b, T y) {
  using L = DivLimits<T>;
  const T A = a * L::up;
  const T Q = div_scaled<T>(a, b, y, L::up);    // RN((a * up) / b): the fast form is exact here
  T res = Q * L::dn;                            // exact if |Q| >= qmin, else RN onto the subnormal grid
  const T diff = dfma<T>(-res, L::up, Q);       // Q - res * up, exact; +-half_step iff Q is a grid midpoint
  const T R = dfma<T>(-b, Q, A);                // exact remainder: true quotient - Q = R / b
  const bool fix = dabs<T>(diff) == L::half_step && R != (T)0 && ((R > (T)0) == (b > (T)0)) == (diff > (T)0);
  const T adj = res + (diff > (T)0 ? L::denorm_min : -L::denorm_min);
  res = fix ? adj : res;
  return a != (T)0 ? res : a * y;
}

template <typename T, bool SMALL_B = false>
__device__ __forceinline__ T div_by_const(T a, T b, T y /* = 1 / b */) {
  using L = DivLimits<T>;
  const T q = a * y;
  const T r = dfma<T>(-b, q, a);
  T res = dfma<T>(r, y, q);
  const T aa = dabs<T>(a);
  if (aa < L::lo) {
    res = q;                                    // a == 0 (the common case of the transport's flux differences): the signed zero of the quotient
    if (a != (T)0) res = div_tiny<T>(a, b, y);
  } else if (SMALL_B && aa > L::hi) {
    res = q;                                    // infinite a: the infinity a * y
    if (aa < L::inf) res = div_scaled<T>(a, b, y, L::dn) * L::up;
  }
  return res;
}

// fp32: |a| in (0, lo) -- a non-zero numerator below the fast window -- by one integer test on the bit
// pattern.  Exact zeros (whole regions before the pressure front arrives, or away from the interface)
// and NaNs are NOT in it: the fast form already returns the signed zero of a * y (or the NaN).  Measured
// (tools/probes/README.md, round 3): -0.7 % on the 2048^2 fp32 bubble step; the fp64 analogue (two
// dword tests instead of one half-rate v_cmp_f64 per quotient) is 3-6 % SLOWER on k_jacobi_tb<double>,
// so fp64 keeps the floating-point test with a second-level zero check.
__device__ __forceinline__ bool tiny_nonzero(float a) {
  constexpr unsigned LO_BITS = __builtin_bit_cast(unsigned, DivLimits<float>::lo);
  return ((__float_as_uint(a) & 0x7fffffffu) - 1u) < (LO_BITS - 1u);
}

// V quotients with ONE branch: all fast forms first (their instruction streams interleave), then a
// single test whether any numerator left the fast window (tiny, zero, NaN), and only then the
// full routine.  With a branch per quotient the compiler cannot overlap the dependent fma chains
// of a lane's V cells.
struct ColdFlag {   // "the tiny / huge tier ran": a register the caller keeps ...
  int* p;
  __device__ __forceinline__ void operator()() const { if (p) *p = 1; }
};
struct ColdFlagLds {   // ... or a word in LDS, which costs a store where the tier runs and nothing where it does not
  typedef __attribute__((address_space(3))) int lds_int;
  lds_int* p;
  __device__ __forceinline__ explicit ColdFlagLds(int* q) : p((lds_int*)q) {}
  __device__ __forceinline__ void operator()() const { *p = 1; }
};
template <typename T, int V, bool SMALL_B = false, typename Cold>
__device__ __forceinline__ void div_by_const_v(T (&res)[V], const T (&a)[V], const T (&b)[V], const T (&y)[V], const Cold& cold);
template <typename T, int V, bool SMALL_B = false>
__device__ __forceinline__ void div_by_const_v(T (&res)[V], const T (&a)[V], const T (&b)[V], const T (&y)[V],
                                               int* cold = nullptr /* set to 1 when the tiny / huge tier ran */) {
  div_by_const_v<T, V, SMALL_B, ColdFlag>(res, a, b, y, ColdFlag{cold});
}
template <typename T, int V, bool SMALL_B, typename Cold>
__device__ __forceinline__ void div_by_const_v(T (&res)[V], const T (&a)[V], const T (&b)[V], const T (&y)[V], const Cold& cold) {
  bool odd = false;
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int q = 0; q < V; ++q) {
      const T q0 = a[q] * y[q];
      res[q] = dfma<T>(dfma<T>(-b[q], q0, a[q]), y[q], q0);
      odd = odd || tiny_nonzero(a[q]) || (SMALL_B && !(dabs<T>(a[q]) <= DivLimits<T>::hi));
    }
    if (odd) {   // (lanes holding a tiny non-zero or a huge / non-finite numerator only)
#pragma unroll
      for (int q = 0; q < V; ++q) {
        if constexpr (SMALL_B) res[q] = div_by_const<T, true>(a[q], b[q], y[q]);
        else {
          res[q] = dabs<T>(a[q]) < DivLimits<T>::lo ? div_tiny<T>(a[q], b[q], y[q]) : res[q];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      cold();
    }
  } else {
#pragma unroll
    for (int q = 0; q < V; ++q) {
      const T q0 = a[q] * y[q];
      res[q] = dfma<T>(dfma<T>(-b[q], q0, a[q]), y[q], q0);
      const T aa = dabs<T>(a[q]);
      odd = odd || !(aa >= DivLimits<T>::lo) || (SMALL_B && !(aa <= DivLimits<T>::hi));
    }
    if (odd) {
      // exact zeros are already right: the fast form returns the signed zero of a * y
      bool nonzero = false;
#pragma unroll
      for (int q = 0; q < V; ++q) {
        const T aa = dabs<T>(a[q]);
        nonzero = nonzero || (a[q] != (T)0 && (!(aa >= DivLimits<T>::lo) || (SMALL_B && !(aa <= DivLimits<T>::hi))));
      }
      if (nonzero) {
#pragma unroll
        for (int q = 0; q < V; ++q) {
          if constexpr (SMALL_B) res[q] = div_by_const<T, true>(a[q], b[q], y[q]);
          else {
            res[q] = dabs<T>(a[q]) < DivLimits<T>::lo ? div_tiny<T>(a[q], b[q], y[q]) : res[q];
            __builtin_amdgcn_sched_barrier(0);   // one cell after the other: interleaved, the cells' temporaries cost k_jacobi_pair its fourth wave per SIMD (127 -> 129 VGPRs)
          }
        }
        cold();
      }
    }
  }
}
// a / b for a numerator known to lie inside the fast window (e.g. a density): no test at all
template <typename T>
__device__ __forceinline__ T div_by_const_inrange(T a, T b, T y) {
  const T q0 = a * y;
  return dfma<T>(dfma<T>(-b, q0, a), y, q0);
}

template <int N> struct IC { static constexpr int value = N; };

}  // namespace vof
