// vof2d_kernels.h -- hand-written gfx950 kernels for the 2-D VOF hot path.
//
// Every kernel cites the reference lines (/root/reference/2dvof.py) it
// replaces.  Arithmetic follows the reference's Python expression order,
// left to right, with FMA contraction disabled (-ffp-contract=off), so the
// results equal the CPU oracle's value for value.
//
// All stencil kernels share one decomposition (vof2d_device.h): a wave owns
// 64*V contiguous columns and marches along i with a register window.
#pragma once
#include "vof2d_device.h"

namespace vof {

// ------------------------------------------------------------------ helpers
// Diagnostic build only (-DVOF_WAVE_TIMES, tools/wave_balance.py): every wave of the selected kernel
// records when it started and ended (s_memrealtime, 100 MHz), which shows how evenly a launch's
// waves finish.  The product build compiles WaveTimer to nothing.
#ifdef VOF_WAVE_TIMES
__device__ unsigned long long* vof_wave_times = nullptr;  // [2 * wave] = start, [2 * wave + 1] = end
__device__ int vof_wave_kid = -1;
__device__ unsigned int vof_wave_cap = 0;
struct WaveTimer {
  unsigned long long t0;
  unsigned int wave;
  bool on;
  __device__ __forceinline__ WaveTimer(int kid) {
    on = vof_wave_times != nullptr && vof_wave_kid == kid;
    wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    t0 = on ? wall_clock64() : 0ull;
  }
  __device__ __forceinline__ ~WaveTimer() {
    if (on && (threadIdx.x & 63) == 0 && wave < vof_wave_cap) {
      vof_wave_times[2 * wave] = t0;
      vof_wave_times[2 * wave + 1] = wall_clock64();
    }
  }
};
#else
struct WaveTimer { __device__ __forceinline__ WaveTimer(int) {} };
#endif
enum : int { WT_MOMENTUM = 0, WT_JACOBI_TB = 3, WT_FCT_X = 5, WT_FCT_Y = 6, WT_JACOBI = 2, WT_TRANSPORT = 12 };  // = KernelId of the runtime
template <typename T, int V>
struct Row {  // one row of a wave tile as seen by a lane: j0-1 | j0..j0+V-1 | j0+V
  T l;
  T c[V];
  T r;
};

template <typename T, int V>
__device__ __forceinline__ void load_c(T (&c)[V], const T* __restrict__ p) {
#if defined(VOF_STREAMING) && VOF_STREAMING >= 2   // experiment: every tile load nontemporal
  typedef T vec_t __attribute__((ext_vector_type(V)));
  vec_t k = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(p));
#pragma unroll
  for (int q = 0; q < V; ++q) c[q] = k[q];
#else
  Pack<T, V> k = *reinterpret_cast<const Pack<T, V>*>(p);
#pragma unroll
  for (int q = 0; q < V; ++q) c[q] = k.v[q];
#endif
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
#ifndef VOF_STREAMING
#define VOF_STREAMING 1
#endif
template <typename T, int V>
__device__ __forceinline__ void load_s(T (&c)[V], const T* __restrict__ p) {
  if constexpr (VOF_STREAMING) load_c_nt<T, V>(c, p); else load_c<T, V>(c, p);
}
template <typename T, int V>
__device__ __forceinline__ void store_s(T* __restrict__ p, const T (&c)[V], int j0, int jlo, int jhi) {
  if constexpr (VOF_STREAMING) store_c_nt<T, V>(p, c, j0, jlo, jhi); else store_c<T, V>(p, c, j0, jlo, jhi);
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
//   * tiny a (the decaying front of the Jacobi iteration walks through 1e-280 ... 4.9e-324 on its
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
  static constexpr double lo = 1e-280, hi = 1e280, up = 0x1p+256, dn = 0x1p-256, qmin = 0x1p-766 /* 2^-1022 * up */,
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

template <typename T, bool SMALL_B = false>
__device__ __forceinline__ T div_by_const(T a, T b, T y /* = 1 / b */) {
  using L = DivLimits<T>;
  const T q = a * y;
  const T r = dfma<T>(-b, q, a);
  T res = dfma<T>(r, y, q);
  const T aa = dabs<T>(a);
  if (aa < L::lo) {
    res = q;                                    // a == 0: signed zero of the quotient
    if (a != (T)0) {
      const T Q = div_scaled<T>(a, b, y, L::up);
      res = Q * L::dn;                          // exact if |Q| >= qmin, else RN onto the subnormal grid
      if (!(dabs<T>(Q) >= L::qmin)) {
        const T diff = Q - res * L::up;         // exact; +-half_step iff Q is a grid midpoint
        const T R = dfma<T>(-b, Q, a * L::up);  // exact remainder: true quotient - Q = R / b
        if (dabs<T>(diff) == L::half_step && R != (T)0 && ((R > (T)0) == (b > (T)0)) == (diff > (T)0))
          res += diff > (T)0 ? L::denorm_min : -L::denorm_min;
      }
    }
  } else if (SMALL_B && aa > L::hi) {
    res = q;                                    // infinite a: the infinity a * y
    if (aa < L::inf) res = div_scaled<T>(a, b, y, L::dn) * L::up;
  }
  return res;
}

// V quotients with ONE branch: all fast forms first (their instruction streams interleave), then a
// single test whether any numerator left the fast window (tiny, zero, NaN), and only then the
// full routine.  With a branch per quotient the compiler cannot overlap the dependent fma chains
// of a lane's V cells.
template <typename T, int V, bool SMALL_B = false>
__device__ __forceinline__ void div_by_const_v(T (&res)[V], const T (&a)[V], const T (&b)[V], const T (&y)[V],
                                               int* cold = nullptr /* set to 1 when the tiny / huge tier ran */) {
  bool odd = false;
#pragma unroll
  for (int q = 0; q < V; ++q) {
    const T q0 = a[q] * y[q];
    res[q] = dfma<T>(dfma<T>(-b[q], q0, a[q]), y[q], q0);
    const T aa = dabs<T>(a[q]);
    odd = odd || !(aa >= DivLimits<T>::lo) || (SMALL_B && !(aa <= DivLimits<T>::hi));
  }
  if (odd) {
    // exact zeros (whole regions before the pressure front arrives, or away from the interface)
    // are already right: the fast form returns the signed zero of a * y
    bool nonzero = false;
#pragma unroll
    for (int q = 0; q < V; ++q) {
      const T aa = dabs<T>(a[q]);
      nonzero = nonzero || (a[q] != (T)0 && (!(aa >= DivLimits<T>::lo) || (SMALL_B && !(aa <= DivLimits<T>::hi))));
    }
    if (nonzero) {
#pragma unroll
      for (int q = 0; q < V; ++q) res[q] = div_by_const<T, SMALL_B>(a[q], b[q], y[q]);
      if (cold) *cold = 1;
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

// ------------------------------------------------------------------ init
// 2dvof.py:102-134 find_area
template <typename T>
__device__ T find_area(const Consts<T>& c, int i, int j, T cx, T cy, T r) {
  T a;
  T xct = (T)(i - 1) * c.dx + c.half_dx;
  T yct = (T)(j - 1) * c.dy + c.half_dy;
  T xlu = xct - c.half_dx, ylu = yct + c.half_dy;
  T xld = xct - c.half_dx, yld = yct - c.half_dy;
  T xru = xct + c.half_dx, yru = yct + c.half_dy;
  T xrd = xct + c.half_dx, yrd = yct - c.half_dy;
#define VOF_DIST(X, Y) dsqrt<T>(((X) - cx) * ((X) - cx) + ((Y) - cy) * ((Y) - cy))
  T dct = VOF_DIST(xct, yct), dlu = VOF_DIST(xlu, ylu), dld = VOF_DIST(xld, yld), dru = VOF_DIST(xru, yru),
    drd = VOF_DIST(xrd, yrd);
#undef VOF_DIST
  if (dlu > r && dld > r && dru > r && drd > r)
    a = (T)1.0;
  else if (dlu < r && dld < r && dru < r && drd < r)
    a = (T)0.0;
  else {
    a = (T)0.5 + (T)0.5 * (dct - r) / c.sqrt2dx;
    a = var3(a, (T)0, (T)1);
  }
  return a;
}

// node coordinate x[k] of 2dvof.py:43-46: hstack((0, linspace(0, L, n+1), L)).astype(f32)
__device__ __forceinline__ double node_coord(double L, int n, int k, int cast_f32) {
  double v = k == 0 ? 0.0 : (k >= n + 1 ? L : (double)(k - 1) * (L / (double)n));
  if (cast_f32) v = (double)(float)v;
  return v;
}

// 2dvof.py:137-159 set_init_F, all stored cells incl. ghosts; writes F and its sweep twin
template <typename T>
__global__ __launch_bounds__(256) void k_init_F(Geom g, Consts<T> c, T* __restrict__ F, T* __restrict__ F2,
                                                 int ic, double Lx, double Ly, int cast_f32) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = g.row_lo + blockIdx.y;
  if (j > g.ny + 1 || i > g.row_hi) return;
  const size_t o = at(g, i, j);
  T val = F[o];
  if (ic == 1) {
    T xi = (T)node_coord(Lx, g.nx, i, cast_f32), yj = (T)node_coord(Ly, g.ny, j, cast_f32);
    if (xi >= (T)0.0 && xi <= c.ic1_x2 && yj >= (T)0.0 && yj <= c.ic1_y2) val = (T)1.0;
  } else if (ic == 2) {
    val = find_area<T>(c, i, j, c.ic_cx, c.ic2_cy, c.ic_r);
  } else {
    val = (T)1.0 - find_area<T>(c, i, j, c.ic_cx, c.ic3_cy, c.ic_r);
    T yj = (T)node_coord(Ly, g.ny, j, cast_f32);
    if (yj < c.ic3_pool) val = (T)1.0;
  }
  F[o] = val;
  F2[o] = val;
}

// ------------------------------------------------------------------ set_BC
// 2dvof.py:162-189.  One thread per row index (loop 1) and per column index
// (loop 2).  Loop 2 reads are redirected to cells loop 1 does not write, and
// loop 1 skips the cells loop 2 overwrites, so one launch reproduces the
// sequential "loop 1 then loop 2" result (corners take loop-2 values, S11).
// F ghosts are mirrored into the sweep twin F2 (see k_fct_*).
// MASK selects the fields (BC_UV | BC_F | BC_P | BC_RHO): the fused step applies each field's
// boundary condition once, right after the field is final (DESIGN.md "schedule").
enum : int { BC_UV = 1, BC_F = 2, BC_P = 4, BC_RHO = 8, BC_ALL = 7 };
template <typename T, int MASK>
__global__ __launch_bounds__(256) void k_set_bc(Geom g, T* __restrict__ u, T* __restrict__ v, T* __restrict__ F,
                                                 T* __restrict__ F2, T* __restrict__ p, T* __restrict__ rho,
                                                 int r0, int r1) {
  constexpr bool UV = MASK & BC_UV, DF = MASK & BC_F, DP = MASK & BC_P, STORED = MASK & BC_RHO;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int ny = g.ny, nx = g.nx;
  // loop 1: row i, restricted to [r0, r1] (a strip leaves the halo rows of a field whose exchange
  // is in flight to the sender, who ships its rows with their ghost columns)
  const int i = g.row_lo + t;
  if (i <= g.row_hi && i >= r0 && i <= r1) {
    const bool wall_row = (g.wall_lo && i == 1) || (g.wall_hi && i == nx + 1);    // u zeroed by loop 2
    const bool ghost_row = (g.wall_lo && i == 0) || (g.wall_hi && i == nx + 1);   // F,p,v,rho from loop 2
    const size_t a0 = at(g, i, 0), a1 = at(g, i, 1), b0 = at(g, i, ny), b1 = at(g, i, ny + 1);
    if (UV && !wall_row) {
      u[a0] = u[a1];
      u[b1] = u[b0];
    }
    if (!ghost_row) {
      if (UV) {
        v[a1] = (T)0;
        v[b1] = (T)0;
      }
      if (DF) {
        T f0 = F[a1], f1 = F[b0];
        F[a0] = f0; F[b1] = f1;
        F2[a0] = f0; F2[b1] = f1;
      }
      if (DP) {
        p[a0] = p[a1];
        p[b1] = p[b0];
      }
      if (STORED) {
        rho[a0] = rho[a1];
        rho[b1] = rho[b0];
      }
    }
  }
  // loop 2: column j
  const int j = t;
  if (j <= ny + 1) {
    const int jj = j == 0 ? 1 : (j == ny + 1 ? ny : j);  // value loop 1 leaves at column j
    const bool vz = (j == 1 || j == ny + 1);             // loop 1 zeroed v there
    if (g.wall_lo) {
      if (UV) {
        u[at(g, 1, j)] = (T)0;
        v[at(g, 0, j)] = vz ? (T)0 : v[at(g, 1, j)];
      }
      if (DF) {
        T f = F[at(g, 1, jj)];
        F[at(g, 0, j)] = f;
        F2[at(g, 0, j)] = f;
      }
      if (DP) p[at(g, 0, j)] = p[at(g, 1, jj)];
      if (STORED) rho[at(g, 0, j)] = rho[at(g, 1, jj)];
    }
    if (g.wall_hi) {
      if (UV) {
        u[at(g, nx + 1, j)] = (T)0;
        v[at(g, nx + 1, j)] = vz ? (T)0 : v[at(g, nx, j)];
      }
      if (DF) {
        T f = F[at(g, nx, jj)];
        F[at(g, nx + 1, j)] = f;
        F2[at(g, nx + 1, j)] = f;
      }
      if (DP) p[at(g, nx + 1, j)] = p[at(g, nx, jj)];
      if (STORED) rho[at(g, nx + 1, j)] = rho[at(g, nx, jj)];
    }
  }
}


// ghost columns of one F buffer for rows [r0, r1]: the F part of set_BC's loop 1 (:162-174) for the
// edge bands of a strip, whose final F leaves for the neighbour before the rest of the rows exist
template <typename T>
__global__ __launch_bounds__(256) void k_bc_F_cols(Geom g, T* __restrict__ F, int r0, int r1) {
  const int i = r0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (i > r1) return;
  F[at(g, i, 0)] = F[at(g, i, 1)];
  F[at(g, i, g.ny + 1)] = F[at(g, i, g.ny)];
}

// ------------------------------------------------------------------ cal_nu_rho
// 2dvof.py:198-203 (verb only: the fused step recomputes rho/nu from F in place)
template <typename T>
__global__ __launch_bounds__(256) void k_nu_rho(Geom g, Consts<T> c, const T* __restrict__ F, T* __restrict__ rho,
                                                 T* __restrict__ nu) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = g.row_lo + blockIdx.y;
  if (j > g.ny + 1 || i > g.row_hi) return;
  const size_t o = at(g, i, j);
  T f = F[o];
  rho[o] = rho_of(c, f);
  nu[o] = nu_of(c, f);
}

// 2dvof.py:452-455 post_process_f on all stored cells (verb); keeps the twin in sync
template <typename T>
__global__ __launch_bounds__(256) void k_post(Geom g, T* __restrict__ F, T* __restrict__ F2) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = g.row_lo + blockIdx.y;
  if (j > g.ny + 1 || i > g.row_hi) return;
  const size_t o = at(g, i, j);
  T f = var3(F[o], (T)0, (T)1);
  F[o] = f;
  F2[o] = f;
}

// ------------------------------------------------------------------ visualisation fields
// 2dvof.py:458-485 get_vof_field / get_u_field / get_v_field / get_vnorm_field: the (2nx, 2ny)
// image rgb_buf[I] = field[I // r] (r = 2), i.e. the *stored* entries [0, nx) x [0, ny) -- ghost
// index 0 included, nx and nx+1 not -- each repeated 2 x 2.  mode 0: F; 1: u / (Lx/0.2);
// 2: v / (Ly/0.2); 3: sqrt(u^2 + v^2) / (Ly/0.2).  img is dense, row-major (2nx, 2ny).
template <typename T>
__global__ __launch_bounds__(256) void k_vis_field(Geom g, const T* __restrict__ F, const T* __restrict__ u,
                                                    const T* __restrict__ v, T* __restrict__ img, int mode,
                                                    T umax, T vmax) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;  // image column index (y)
  const int a = blockIdx.y;                             // image row index (x)
  if (b >= 2 * g.ny || a >= 2 * g.nx) return;
  const size_t o = at(g, a / 2, b / 2);
  T val;
  if (mode == 0) val = F[o];
  else if (mode == 1) val = u[o] / umax;
  else if (mode == 2) val = v[o] / vmax;
  else val = dsqrt<T>(u[o] * u[o] + v[o] * v[o]) / vmax;
  img[(size_t)a * (size_t)(2 * g.ny) + b] = val;
}

// 2dvof.py:488-492 interp_velocity: V[i,j] = ((u[i,j]+u[i+1,j])/2, (v[i,j]+v[i,j+1])/2) for
// i in [1, nx+1], j in [1, ny].  At i = nx+1 the reference indexes u[nx+2, j], one row past the
// field (undefined in Taichi's release mode); it reads as 0 here.  out is dense (nx+2, ny+2, 2),
// entries outside the loop range stay 0 like the zero-initialised ti.Vector.field.
template <typename T>
__global__ __launch_bounds__(256) void k_interp_velocity(Geom g, const T* __restrict__ u, const T* __restrict__ v,
                                                          T* __restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (j > g.ny + 1 || i > g.nx + 1) return;
  T vx = (T)0, vy = (T)0;
  if (i >= 1 && j >= 1 && j <= g.ny) {
    const T unext = i + 1 <= g.nx + 1 ? u[at(g, i + 1, j)] : (T)0;
    vx = (u[at(g, i, j)] + unext) / (T)2;
    vy = (v[at(g, i, j)] + v[at(g, i, j + 1)]) / (T)2;
  }
  const size_t o = ((size_t)i * (size_t)(g.ny + 2) + j) * 2;
  out[o] = vx;
  out[o + 1] = vy;
}

// ------------------------------------------------------------------ normals
// 2dvof.py:285-306 get_normal_young loop 1: F (3x3) -> mx, my on interior rows.
template <typename T, int V>
__global__ __launch_bounds__(256) void k_normals(Geom g, Consts<T> c, const T* __restrict__ F, T* __restrict__ mx,
                                                  T* __restrict__ my, int R) {
  int j0, ra, rb;
  if (!wave_tile<V>(g, g.ilo, g.ihi, R, j0, ra, rb)) return;
  const T cxn = c.nrm_x, cyn = c.nrm_y;
  size_t o = at(g, ra, j0);
  Row<T, V> m, z, p;  // rows i-1, i, i+1
  load_row<T, V>(m, F + o - g.pitch);
  load_row<T, V>(z, F + o);
  for (int i = ra; i <= rb; ++i, o += g.pitch) {
    load_row<T, V>(p, F + o + g.pitch);
    T ox[V], oy[V];
#pragma unroll
    for (int q = 0; q < V; ++q) {
      const T Fmm = left_of(m, q), Fm0 = m.c[q], Fmp = right_of(m, q);
      const T F0m = left_of(z, q), F00 = z.c[q], F0p = right_of(z, q);
      const T Fpm = left_of(p, q), Fp0 = p.c[q], Fpp = right_of(p, q);
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
        ox[q] = mxsum;
        oy[q] = mysum;
      } else {
        T magnitude = dsqrt<T>(mxsum * mxsum + mysum * mysum);
        ox[q] = mxsum / magnitude;
        oy[q] = mysum / magnitude;
      }
    }
    store_c<T, V>(mx + o, ox, j0, 1, g.ny);
    store_c<T, V>(my + o, oy, j0, 1, g.ny);
    m = z;
    z = p;
  }
}

// 2dvof.py:307-309 get_normal_young loop 2: kappa from mx (i+-1) and my (j+-1)
template <typename T, int V>
__global__ __launch_bounds__(256) void k_kappa(Geom g, Consts<T> c, const T* __restrict__ mx,
                                                const T* __restrict__ my, T* __restrict__ kappa, int R) {
  int j0, ra, rb;
  if (!wave_tile<V>(g, g.ilo, g.ihi, R, j0, ra, rb)) return;
  size_t o = at(g, ra, j0);
  T xm[V], xz[V], xp[V];
  load_c<T, V>(xm, mx + o - g.pitch);
  load_c<T, V>(xz, mx + o);
  for (int i = ra; i <= rb; ++i, o += g.pitch) {
    load_c<T, V>(xp, mx + o + g.pitch);
    Row<T, V> y;
    load_row<T, V>(y, my + o);
    T k[V];
#pragma unroll
    for (int q = 0; q < V; ++q)
      k[q] = -(c.kap_x * (xp[q] - xm[q]) + c.kap_y * (right_of(y, q) - left_of(y, q)));
    store_c<T, V>(kappa + o, k, j0, 1, g.ny);
#pragma unroll
    for (int q = 0; q < V; ++q) {
      xm[q] = xz[q];
      xz[q] = xp[q];
    }
  }
}

// ------------------------------------------------------------------ predictor
// 2dvof.py:206-233 advect_upwind: u*, v* from u, v, kappa, F (rho, nu).
// STORED: read the rho / nu arrays written by cal_nu_rho (verb semantics);
// otherwise recompute them from F per cell (identical values: rho[i,j] is a
// pure function of F[i,j] and F is unchanged since cal_nu_rho, 2dvof.py:513-517).
template <typename T, int V, bool STORED>
__global__ __launch_bounds__(256) void k_predictor(Geom g, Consts<T> c, const T* __restrict__ u,
                                                    const T* __restrict__ v, const T* __restrict__ kappa,
                                                    const T* __restrict__ F, const T* __restrict__ rho,
                                                    const T* __restrict__ nu, T* __restrict__ us,
                                                    T* __restrict__ vs, int R) {
  int j0, ra, rb;
  if (!wave_tile<V>(g, g.ilo, g.ihi, R, j0, ra, rb)) return;
  const T dt = c.dt, dxi = c.dxi, dyi = c.dyi, dxi2 = c.dxi2, dyi2 = c.dyi2;
  size_t o = at(g, ra, j0);
  Row<T, V> um, uz, up, vm, vz, vp;
  T km[V], Fm[V], rm_[V];
  load_row<T, V>(um, u + o - g.pitch);
  load_row<T, V>(uz, u + o);
  load_row<T, V>(vm, v + o - g.pitch);
  load_row<T, V>(vz, v + o);
  load_c<T, V>(km, kappa + o - g.pitch);
  load_c<T, V>(Fm, F + o - g.pitch);
  if (STORED) load_c<T, V>(rm_, rho + o - g.pitch);
  for (int i = ra; i <= rb; ++i, o += g.pitch) {
    load_row<T, V>(up, u + o + g.pitch);
    load_row<T, V>(vp, v + o + g.pitch);
    Row<T, V> kz, Fz, rz;
    T nz[V];
    load_c<T, V>(kz.c, kappa + o);
    kz.l = kappa[o - 1];
    load_c<T, V>(Fz.c, F + o);
    Fz.l = F[o - 1];
    if (STORED) {
      load_c<T, V>(rz.c, rho + o);
      rz.l = rho[o - 1];
      load_c<T, V>(nz, nu + o);
    }
    T ou[V], ov[V];
#pragma unroll
    for (int q = 0; q < V; ++q) {
      const T u00 = uz.c[q], um0 = um.c[q], up0 = up.c[q], u0m = left_of(uz, q), u0p = right_of(uz, q);
      const T upm = left_of(up, q);
      const T v00 = vz.c[q], vm0 = vm.c[q], vp0 = vp.c[q], v0m = left_of(vz, q), v0p = right_of(vz, q);
      const T vmp = right_of(vm, q);
      const T F00 = Fz.c[q], Fm0 = Fm[q], F0m = left_of(Fz, q);
      const T k00 = kz.c[q], km0 = km[q], k0m = left_of(kz, q);
      T rho00, rhom0, rho0m, nu00;
      if (STORED) {
        rho00 = rz.c[q]; rhom0 = rm_[q]; rho0m = left_of(rz, q); nu00 = nz[q];
      } else {
        rho00 = rho_of(c, F00); rhom0 = rho_of(c, Fm0); rho0m = rho_of(c, F0m); nu00 = nu_of(c, F00);
      }
      {  // :208-220
        T v_here = (T)0.25 * (vm0 + vmp + v00 + v0p);
        T dudx = u00 > 0 ? (u00 - um0) * dxi : (up0 - u00) * dxi;
        T dudy = v_here > 0 ? (u00 - u0m) * dyi : (u0p - u00) * dyi;
        T kappa_ave = (k00 + km0) / (T)2.0;
        T fx_kappa = -c.sigma * (F00 - Fm0) * kappa_ave / c.dx;
        ou[q] = (u00 + dt * (nu00 * (um0 - (T)2 * u00 + up0) * dxi2 + nu00 * (u0m - (T)2 * u00 + u0p) * dyi2 -
                             u00 * dudx - v_here * dudy + c.gx + fx_kappa * (T)2 / (rho00 + rhom0)));
      }
      {  // :221-233
        T u_here = (T)0.25 * (u0m + u00 + upm + up0);
        T dvdx = u_here > 0 ? (v00 - vm0) * dxi : (vp0 - v00) * dxi;
        T dvdy = v00 > 0 ? (v00 - v0m) * dyi : (v0p - v00) * dyi;
        T kappa_ave = (k00 + k0m) / (T)2.0;
        T fy_kappa = -c.sigma * (F00 - F0m) * kappa_ave / c.dy;
        ov[q] = (v00 + dt * (nu00 * (vm0 - (T)2 * v00 + vp0) * dxi2 + nu00 * (v0m - (T)2 * v00 + v0p) * dyi2 -
                             u_here * dvdx - v00 * dvdy + c.gy + fy_kappa * (T)2 / (rho00 + rho0m)));
      }
    }
    if (i >= 2) store_c<T, V>(us + o, ou, j0, 1, g.ny);  // i in [imin+1, imax]
    store_c<T, V>(vs + o, ov, j0, 2, g.ny);               // j in [jmin+1, jmax]
    um = uz; uz = up; vm = vz; vz = vp;
#pragma unroll
    for (int q = 0; q < V; ++q) {
      km[q] = kz.c[q];
      Fm[q] = Fz.c[q];
      if (STORED) rm_[q] = rz.c[q];
    }
  }
}

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
constexpr int TB_SLOW10 = 20;         // cost of a band row in tenths of an ordinary row
struct TbPlan {
  unsigned long long* masks;          // nullptr: no plan (uniform layout)
  unsigned long long* plan;
  int ntt, R, waves, par;             // tile columns (<= TB_COLS), uniform chunk length, waves of a launch, istep & 1
};
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
struct TbPlanShared {
  unsigned prefix[TB_COLS][TB_BANDS + 1];      // prefix[j][b] = cost of rows [0, b * bh) of tile column j, in tenths of a row
  unsigned long long band[TB_BANDS][TB_COLS / 64];   // the reported (band, column) bits
  int first[TB_COLS + 1];                      // first wave of column j; first[TB_COLS] = planned waves
  int n[TB_COLS];                              // chunks of column j
};
__device__ __forceinline__ bool tb_bit(const TbPlanShared& sh, int b, int j) { return ((sh.band[b][j >> 6] >> (j & 63)) & 1ull) != 0ull; }
// row position (0 .. rows) where the cumulative cost of column j reaches T
__device__ __forceinline__ int tb_pos(const TbPlanShared& sh, int j, unsigned T, int rows, int bh) {
  int lo = 0, hi = TB_BANDS;           // largest b with prefix[j][b] <= T
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (sh.prefix[j][mid] <= T) lo = mid; else hi = mid;
  }
  const bool slow = tb_bit(sh, lo, j);
  const unsigned rest = T - sh.prefix[j][lo];
  int pos = lo * bh + (int)(slow ? rest / (unsigned)TB_SLOW10 : rest / 10u);
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
    if (lane == 0) tp.plan[0] = any ? 1ull : 0ull;
  }
  if (!any) return;                             // block-uniform
  __syncthreads();
  if (t < 64) {   // wave 0: per column (lane j and j + 64), the cost prefix over the bands and the number of chunks
    unsigned cost[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      const int j = lane + 64 * c;
      unsigned acc = 0;
      for (int b = 0; b < TB_BANDS; ++b) {
        sh.prefix[j][b] = acc;
        const int r0 = b * bh, r1 = r0 + bh < rows ? r0 + bh : rows;
        if (r1 > r0) acc += (unsigned)(r1 - r0) * (tb_bit(sh, b, j) ? (unsigned)TB_SLOW10 : 10u);
      }
      sh.prefix[j][TB_BANDS] = acc;
      cost[c] = j < tp.ntt ? acc : 0u;
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
      const unsigned cost = sh.prefix[j][TB_BANDS];
      const int a = k == 0 ? 0 : tb_pos(sh, j, (unsigned)(((unsigned long long)cost * (unsigned)k) / (unsigned)n), rows, bh);
      const int b = k == n - 1 ? rows : tb_pos(sh, j, (unsigned)(((unsigned long long)cost * (unsigned)(k + 1)) / (unsigned)n), rows, bh);
      e = plan_pack(j, g.ilo + a, g.ilo + b - 1);   // (b == a: an empty chunk, the wave returns at once)
    }
    tp.plan[1 + w] = e;
  }
}

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

template <typename T, int V>
__global__ __launch_bounds__(256) void k_momentum(Geom g, Consts<T> c, const T* __restrict__ F,
                                                   const T* __restrict__ u, const T* __restrict__ v,
                                                   T* __restrict__ us, T* __restrict__ vs, T* __restrict__ rhs,
                                                   int R, int ntt, int virt, TbPlan tp) {
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
  constexpr int H = ((2 + V - 1) / V) * V;
  WaveTimer wt_(WT_MOMENTUM);
  constexpr int STRIDE = W - 2 * H;
  const int wave = ((int)blockIdx.x - plan_blocks) * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int tj = wave % ntt, ch = wave / ntt;
  const int c0 = 1 - H + tj * STRIDE;
  const int j0 = c0 + lane * V;
  const int ra = g.ilo + ch * R;
  if (ra > g.ihi) return;  // wave-uniform
  const int rb = ra + R - 1 < g.ihi ? ra + R - 1 : g.ihi;
  const int ny = g.ny, ilo = g.ilo, ihi = g.ihi;
  const int jlo = c0 + H > 1 ? c0 + H : 1;
  const int jhi = c0 + W - H - 1 < ny ? c0 + W - H - 1 : ny;
  const T dt = c.dt, dxi = c.dxi, dyi = c.dyi, dxi2 = c.dxi2, dyi2 = c.dyi2;
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
  // windows; index names are relative to the newest F row r of the current iteration
  Row<T, V> F2, F1;            // F rows r-2, r-1 (become r-3.. after the shift)
  T F3c[V];                    // F row r-3, centre columns
  Row<T, V> u3, u2, v3, v2;    // u, v rows r-3, r-2 (row r-1 is loaded in the iteration)
  T mx2[V], mx3[V], my2[V];    // mx rows r-2, r-3; my row r-2
  T k3[V];                     // kappa row r-3
  T us3[V], vs3[V];            // u*, v* row r-3
  T rho3[V];                   // rho(F) row r-3 (rho is a pure function of F[i,j], :201-202)
  const int r0 = ra - 1, r1 = rb + 3;
  load_F(F2, r0 - 2);
  load_F(F1, r0 - 1);
  // u, v rows below ra-1 are never used by a stored value (the first stored u*, v* row is ra, which
  // reads rows ra-1 .. ra+1; F needs ra-3 .. for the normals behind kappa): not loaded, the window
  // starts from zeros (three row loads per array and chunk less)
  auto zero_row = [](Row<T, V>& w) {
    w.l = w.r = (T)0;
#pragma unroll
    for (int q = 0; q < V; ++q) w.c[q] = (T)0;
  };
  zero_row(u3); zero_row(u2); zero_row(v3); zero_row(v2);
  if (edge_cols) {
    mirror_ghost_cols<T, V>(F2, j0, ny);
    mirror_ghost_cols<T, V>(F1, j0, ny);
  }
#pragma unroll
  for (int q = 0; q < V; ++q) F3c[q] = mx2[q] = mx3[q] = my2[q] = k3[q] = us3[q] = vs3[q] = rho3[q] = (T)0;
  bool flat2 = row_flat<T, V>(F2), flat1 = row_flat<T, V>(F1), flat0;  // rows r-2, r-1, r all-equal tests
  Row<T, V> Fn, un, vn;  // prefetched: F row r, u / v row r-1
  load_F(Fn, r0);
  zero_row(un); zero_row(vn);   // (row ra-2: unused, see above)
  for (int r = r0; r <= r1; ++r) {
    Row<T, V> F0 = Fn, u1 = un;
    const Row<T, V> v1 = vn;
    if (r < r1) {
      load_F(Fn, r + 1);
      load_u(un, r);
      load_v(vn, r);
    }
    if (edge_cols) {   // (after the prefetch has been issued)
      mirror_ghost_cols<T, V>(F0, j0, ny);
      mirror_ghost_cols<T, V>(u1, j0, ny);
    }
    // ---- N: normals of row r-1 (:285-306)
    const bool okN = (r - 1) >= ilo && (r - 1) <= ihi;
    T mx1[V], my1[V];
    // Away from the interface all 3 x (V+2) values of F a lane sees are equal; every corner
    // difference of :287-294 is then an exact zero and (mx, my) = (0, 0).  When that holds for the
    // whole wave the stage is skipped (flatF[k] caches the per-row test, one row is new per step).
    flat0 = row_flat<T, V>(F0);
    const bool flat = flat2 && flat1 && flat0 && F2.c[0] == F1.c[0] && F1.c[0] == F0.c[0];
    if (__all(flat)) {
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
    const bool okK = (r - 2) >= ilo && (r - 2) <= ihi;
    const T myl = lane_up(my2[V - 1]), myr = lane_dn(my2[0]);
    T k2[V];
#pragma unroll
    for (int q = 0; q < V; ++q) {
      const T yr = q == V - 1 ? myr : my2[q + 1], yl = q == 0 ? myl : my2[q - 1];
      const T kk = -(c.kap_x * (mx1[q] - mx3[q]) + c.kap_y * (yr - yl));
      k2[q] = (okK && dom[q]) ? kk : (T)0;
    }
    // ---- P: u*, v* of row i = r-2 (:206-233)
    const int i = r - 2;
    const bool okP = i >= ilo && i <= ihi;
    const T kl = lane_up(k2[V - 1]);
    T us2[V], vs2[V], rho2[V];
#pragma unroll
    for (int q = 0; q < V; ++q) rho2[q] = rho_of(c, F2.c[q]);
    const T rho2l = rho_of(c, F2.l);
    // Surface tension (:213-214, :225-226): force = (-sigma * dF * kappa_ave / dx) * 2 / (rho + rho').
    // Away from the interface dF or kappa_ave is an exact zero and so is the force; one wave-level
    // test covers the 2 V quotient pairs of the lane, and the exact divisions run only behind it.
    T fxf[V], fyf[V];
    bool any_force = false;
#pragma unroll
    for (int q = 0; q < V; ++q) {
      const T F00 = F2.c[q], Fm0 = F3c[q], F0m = left_of(F2, q);
      const T k00 = k2[q], km0 = k3[q], k0m = q == 0 ? kl : k2[q - 1];
      fxf[q] = -c.sigma * (F00 - Fm0) * ((k00 + km0) / (T)2.0);
      fyf[q] = -c.sigma * (F00 - F0m) * ((k00 + k0m) / (T)2.0);
      any_force = any_force || fxf[q] != (T)0 || fyf[q] != (T)0;
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
      const T nu00 = nu_of(c, F2.c[q]);
      T ou, ov;
      {
        T v_here = (T)0.25 * (vm0 + vmp + v00 + v0p);
        T dudx = upwind_diff<T>(u00 > 0, u00, um0, up0) * dxi;      // (u00-um0)*dxi or (up0-u00)*dxi
        T dudy = upwind_diff<T>(v_here > 0, u00, u0m, u0p) * dyi;
        ou = (u00 + dt * (nu00 * (um0 - (T)2 * u00 + up0) * dxi2 + nu00 * (u0m - (T)2 * u00 + u0p) * dyi2 -
                          u00 * dudx - v_here * dudy + c.gx + fxf[q]));
      }
      {
        T u_here = (T)0.25 * (u0m + u00 + upm + up0);
        T dvdx = upwind_diff<T>(u_here > 0, v00, vm0, vp0) * dxi;
        T dvdy = upwind_diff<T>(v00 > 0, v00, v0m, v0p) * dyi;
        ov = (v00 + dt * (nu00 * (vm0 - (T)2 * v00 + vp0) * dxi2 + nu00 * (v0m - (T)2 * v00 + v0p) * dyi2 -
                          u_here * dvdx - v00 * dvdy + c.gy + fyf[q]));
      }
      const int j = j0 + q;
      us2[q] = (okP && i >= 2 && dom[q]) ? ou : (T)0;           // u* exists on i in [2, nx]
      vs2[q] = (okP && j >= 2 && j <= ny) ? ov : (T)0;          // v* exists on j in [2, ny]
    }
    if (i >= ra && i <= rb) {
      if (i >= 2) store_s<T, V>(us + at(g, i, j0), us2, j0, jlo, jhi);
      store_s<T, V>(vs + at(g, i, j0), vs2, j0, jlo > 2 ? jlo : 2, jhi);
    }
    // ---- R: rhs of row r-3 (:239-241)
    const int i3 = r - 3;
    if (i3 >= ra && i3 <= rb) {
      const T vsr = lane_dn(vs3[0]);
      T out[V];
#pragma unroll
      for (int q = 0; q < V; ++q) {
        const T vright = q == V - 1 ? vsr : vs3[q + 1];
        // rho lies in [rho_g, rho_l] (var clamps F, :192-196): always inside the fast window
        out[q] = div_by_const_inrange<T>(rho3[q], c.dt, c.inv_dt) *
                 ((us2[q] - us3[q]) * c.dxi + (vright - vs3[q]) * c.dyi);
      }
      store_s<T, V>(rhs + at(g, i3, j0), out, j0, jlo, jhi);
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
}

// ------------------------------------------------------------------ rhs
// 2dvof.py:239-241, hoisted out of the Jacobi loop (it does not depend on p;
// precedent: cal_velocity_div, diff_vof_replaced.py:277-282).
template <typename T, int V, bool STORED>
__global__ __launch_bounds__(256) void k_rhs(Geom g, Consts<T> c, const T* __restrict__ us,
                                              const T* __restrict__ vs, const T* __restrict__ F,
                                              const T* __restrict__ rho, T* __restrict__ rhs, int R) {
  int j0, ra, rb;
  if (!wave_tile<V>(g, g.ilo, g.ihi, R, j0, ra, rb)) return;
  size_t o = at(g, ra, j0);
  T uz[V], up[V];
  load_c<T, V>(uz, us + o);
  for (int i = ra; i <= rb; ++i, o += g.pitch) {
    load_c<T, V>(up, us + o + g.pitch);
    Row<T, V> vz;
    load_c<T, V>(vz.c, vs + o);
    vz.r = vs[o + V];
    T f[V], out[V];
    load_c<T, V>(f, STORED ? rho + o : F + o);
#pragma unroll
    for (int q = 0; q < V; ++q) {
      T r = STORED ? f[q] : rho_of(c, f[q]);
      out[q] = r / c.dt * ((up[q] - uz[q]) * c.dxi + (right_of(vz, q) - vz.c[q]) * c.dyi);
    }
    store_c<T, V>(rhs + o, out, j0, 1, g.ny);
#pragma unroll
    for (int q = 0; q < V; ++q) uz[q] = up[q];
  }
}

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
#ifndef VOF_TB_MINWAVES
#define VOF_TB_MINWAVES 1
#endif
template <typename T, int V, int TS, bool SQ, bool RESID = false>
__global__ __launch_bounds__(256, VOF_TB_MINWAVES) void k_jacobi_tb(Geom g, Consts<T> c, const T* __restrict__ p,
                                                    const T* __restrict__ rhs, T* __restrict__ pn, int R,
                                                    int ntt, unsigned long long* __restrict__ norm_bits = nullptr,
                                                    TbPlan tp = TbPlan{nullptr, nullptr, 0, 0, 0, 0}) {
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
  const int wave = xcd_contiguous_block(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  // wave -> (tile column tj, rows [ra, rb]): chunks of R rows of every column, or, while the tiny-value
  // front crosses the grid, the equal-cost chunks of the step's plan (tb_make_plan)
  int tj = wave % ntt, ra = g.ilo + (wave / ntt) * R, rb = ra + R - 1;
  bool planned = false;
  if (tp.masks != nullptr) {
    const unsigned long long* pl = tp.plan;
    if (pl[0] != 0ull) {
      const unsigned long long e = pl[1 + wave];    // scalar loads: nobody writes the plan during the launch
      tj = (int)(e & 0xffull);
      ra = (int)((e >> 8) & 0xfffffffull);
      rb = (int)((e >> 36) & 0xfffffffull);
      planned = true;
    }
  }
  tj = __builtin_amdgcn_readfirstlane(tj);
  ra = __builtin_amdgcn_readfirstlane(ra);
  rb = __builtin_amdgcn_readfirstlane(rb);
  if (planned && rb < ra) return;   // an unused wave of the plan
  const int c0 = 1 - H + tj * STRIDE;
  const int j0 = c0 + lane * V;
  if (ra > g.ihi) return;  // wave-uniform
  if (rb > g.ihi) rb = g.ihi;
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
      if (s == 1 && t < t1) {
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
    if (io >= ra && io <= rb) store_s<T, V>(pn + at(g, io, j0), carry, j0, jlo, jhi);
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

// ------------------------------------------------------------------ corrector
// 2dvof.py:269-280 update_uv (+ Courant prints -> counter over owned rows)
template <typename T, int V, bool STORED>
__global__ __launch_bounds__(256) void k_correct(Geom g, Consts<T> c, const T* __restrict__ p,
                                                  const T* __restrict__ F, const T* __restrict__ rho,
                                                  const T* __restrict__ us, const T* __restrict__ vs,
                                                  T* __restrict__ u, T* __restrict__ v, int R,
                                                  unsigned long long* __restrict__ courant) {
  int j0, ra, rb;
  if (!wave_tile<V>(g, g.ilo, g.ihi, R, j0, ra, rb)) return;
  size_t o = at(g, ra, j0);
  T pm[V], rm_[V];
  load_c<T, V>(pm, p + o - g.pitch);
  {
    T f[V];
    load_c<T, V>(f, STORED ? rho + o - g.pitch : F + o - g.pitch);
#pragma unroll
    for (int q = 0; q < V; ++q) rm_[q] = STORED ? f[q] : rho_of(c, f[q]);
  }
  unsigned int viol = 0;
  for (int i = ra; i <= rb; ++i, o += g.pitch) {
    Row<T, V> pz, rz;
    load_c<T, V>(pz.c, p + o);
    pz.l = p[o - 1];
    {
      T f[V];
      load_c<T, V>(f, STORED ? rho + o : F + o);
      T fl = STORED ? rho[o - 1] : F[o - 1];
#pragma unroll
      for (int q = 0; q < V; ++q) rz.c[q] = STORED ? f[q] : rho_of(c, f[q]);
      rz.l = STORED ? fl : rho_of(c, fl);
    }
    T usz[V], vsz[V], ou[V], ov[V];
    load_c<T, V>(usz, us + o);
    load_c<T, V>(vsz, vs + o);
    const bool own = i >= g.own_lo && i <= g.own_hi;
#pragma unroll
    for (int q = 0; q < V; ++q) {
      T r = (rz.c[q] + rm_[q]) * (T)0.5;
      ou[q] = usz[q] - c.dt / r * (pz.c[q] - pm[q]) * c.dxi;
      T r2 = (rz.c[q] + left_of(rz, q)) * (T)0.5;
      ov[q] = vsz[q] - c.dt / r2 * (pz.c[q] - left_of(pz, q)) * c.dyi;
      const int j = j0 + q;
      if (own && j <= g.ny) {
        if (i >= 2 && ou[q] * c.dt > c.cfl_x) viol++;
        if (j >= 2 && ov[q] * c.dt > c.cfl_y) viol++;
      }
    }
    if (i >= 2) store_c<T, V>(u + o, ou, j0, 1, g.ny);
    store_c<T, V>(v + o, ov, j0, 2, g.ny);
#pragma unroll
    for (int q = 0; q < V; ++q) {
      pm[q] = pz.c[q];
      rm_[q] = rz.c[q];
    }
  }
  if (__any(viol != 0)) {
    unsigned int tot = viol;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) tot += __shfl_down(tot, s, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(courant, (unsigned long long)tot);
  }
}

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
template <typename T, int V, bool POST>
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
    td[q] = (j >= 1 && j <= ny) ? fct_ftd<T>(c, Fz[q], L[q], q == V - 1 ? Ln : L[q + 1], dv[q]) : (T)0;
  }
  const T tl = lane_up(td[V - 1]), tr = lane_dn(td[0]);
  T rp[V], rm[V];
#pragma unroll
  for (int q = 0; q < V; ++q) {
    const int j = j0 + q;
    rp[q] = rm[q] = (T)0;
    if (j >= 1 && j <= ny)
      fct_ratios<T>(c, td[q], q == 0 ? tl : td[q - 1], q == V - 1 ? tr : td[q + 1], a[q],
                    q == V - 1 ? an_ : a[q + 1], rp[q], rm[q]);
  }
  const T rpl = lane_up(rp[V - 1]), rml = lane_up(rm[V - 1]);
  T cy[V];
#pragma unroll
  for (int q = 0; q < V; ++q) {
    const int j = j0 + q;  // face j between cells j-1 and j; written for j in [2, ny+1]
    cy[q] = (j >= 2 && j <= ny + 1)
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
  __device__ __forceinline__ void init(const T (&Fm)[V]) {  // Fm = F[row before the first pushed row]
#pragma unroll
    for (int q = 0; q < V; ++q) {
      F1[q] = Fm[q];
      u1[q] = L1[q] = a1[q] = a2[q] = a3[q] = t2[q] = t3[q] = rp3[q] = rm3[q] = c3[q] = (T)0;
      d2[q] = d3[q] = (T)1;
    }
    zrows = 0;
  }
  template <bool POST>
  __device__ __forceinline__ void push(const Consts<T>& c, int r, int ilo, int ihi, const T (&Fr)[V],
                                       const T (&ur)[V], T (&out)[V]) {
    bool rz = true;
#pragma unroll
    for (int q = 0; q < V; ++q) rz = rz && Fr[q] == (T)0;
    zrows = __all(rz) ? zrows + 1 : 0;
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
    } else {
#pragma unroll
      for (int q = 0; q < V; ++q) {
        T Lr, ar;
        fct_face<T>(ur[q], c.dt, F1[q], Fr[q], Lr, ar);
        const int i1 = r - 1;
        T dv1 = c.dxdy - c.dtdy * (ur[q] - u1[q]);
        T tn = (i1 >= ilo && i1 <= ihi) ? fct_ftd<T>(c, F1[q], L1[q], Lr, dv1) : (T)0;
        const int i2 = r - 2;
        T rp2 = (T)0, rm2 = (T)0;
        if (i2 >= ilo && i2 <= ihi) fct_ratios<T>(c, t2[q], t3[q], tn, a2[q], a1[q], rp2, rm2);
        T c2 = (i2 > ilo && i2 <= ihi + 1) ? fct_climit<T>(a2[q], rp3[q], rm3[q], rp2, rm2) : (T)0;
        out[q] = fct_final<T, POST>(c, t3[q], a3[q], c3[q], a2[q], c2, d3[q]);
        F1[q] = Fr[q]; u1[q] = ur[q]; L1[q] = Lr;
        a3[q] = a2[q]; a2[q] = a1[q]; a1[q] = ar;
        t3[q] = t2[q]; t2[q] = tn;
        d3[q] = d2[q]; d2[q] = dv1;
        rp3[q] = rp2; rm3[q] = rm2; c3[q] = c2;
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
    pipe.template push<POST>(c, r, ilo, ihi, Fr, ur, out);
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
  constexpr int W = 64 * V, STRIDE = W - 8;
  WaveTimer wt_(WT_FCT_Y);
  const int wave = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // SGPR: rows are wave-uniform
  const int lane = threadIdx.x & 63;
  const int tj = wave % nty, ch = wave / nty;
  const int c0 = -3 + tj * STRIDE;
  const int j0 = c0 + lane * V;
  const int ra = rfirst + ch * R;
  if (ra > rlast) return;  // wave-uniform
  const int rb = ra + R - 1 < rlast ? ra + R - 1 : rlast;
  const int ny = g.ny;
  const int jlo = c0 + 4 > 1 ? c0 + 4 : 1;
  const int jhi = c0 + W - 5 < ny ? c0 + W - 5 : ny;
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
  constexpr int W = 64 * V, STRIDE = W - 8;
  WaveTimer wt_(WT_TRANSPORT);
  const int wave = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int tj = wave % nty, ch = wave / nty;
  const int c0 = -3 + tj * STRIDE;
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
  const int jlo = c0 + 4 > 1 ? c0 + 4 : 1;
  const int jhi = c0 + W - 5 < ny ? c0 + W - 5 : ny;
  auto rowptr = [&](const T* base, int r) {
    const int rc = r < g.row_lo ? g.row_lo : (r > g.row_hi ? g.row_hi : r);
    return base + at(g, rc, j0);
  };
  FctXPipe<T, V> pipe;
  T p1[V], rho1[V];   // p and rho of the previous row (update_uv's i-1 operands)
  {
    T f1[V];
    load_c<T, V>(f1, rowptr(F, ra - 3));
    load_c<T, V>(p1, rowptr(p, ra - 3));
#pragma unroll
    for (int q = 0; q < V; ++q) rho1[q] = rho_of(c, f1[q]);
    if (YFIRST && ra - 3 >= ilo) {
      // the pipeline's first donor cell is row ra-3 of the y-swept F: sweep that row here (its
      // corrected v needs operands of the same row only)
      T v0[V], fs[V];
      load_s<T, V>(v0, rowptr(vs, ra - 3));
      const T rhol = lane_up(rho1[V - 1]), pl = lane_up(p1[V - 1]);
#pragma unroll
      for (int q = 0; q < V; ++q) {
        const int j = j0 + q;
        const T vn = corrected_velocity<T>(c, v0[q], rho1[q], q == 0 ? rhol : rho1[q - 1], p1[q],
                                           q == 0 ? pl : p1[q - 1], c.dyi);
        v0[q] = (j >= 2 && j <= ny) ? vn : (T)0;
      }
      fct_y_row<T, V, false>(c, j0, ny, f1, v0, fs);
      pipe.init(fs);
    } else {
      pipe.init(f1);
    }
  }
  T v1[V], v2[V], v3[V];  // x first: corrected v of rows r-1, r-2, r-3 (the y sweep trails the pipeline)
#pragma unroll
  for (int q = 0; q < V; ++q) v1[q] = v2[q] = v3[q] = (T)0;
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
    {  // update_uv for row r (:269-280): ur / vr hold u*[r] / v*[r]
      T rhor[V];
#pragma unroll
      for (int q = 0; q < V; ++q) rhor[q] = rho_of(c, Fr[q]);
      const T rhol = lane_up(rhor[V - 1]), pl = lane_up(pr[V - 1]);
      const bool urow = r >= 2 && r <= nx;     // u exists on i in [2, nx]; the walls keep 0
      const bool own = r >= ra && r <= rb;     // rows this chunk stores (and counts)
#pragma unroll
      for (int q = 0; q < V; ++q) {
        const int j = j0 + q;
        const T un = corrected_velocity<T>(c, ur[q], rhor[q], rho1[q], pr[q], p1[q], c.dxi);
        ur[q] = urow ? un : (T)0;
        const T vn = corrected_velocity<T>(c, vr[q], rhor[q], q == 0 ? rhol : rhor[q - 1], pr[q],
                                           q == 0 ? pl : pr[q - 1], c.dyi);
        vr[q] = (j >= 2 && j <= ny) ? vn : (T)0;   // v exists on j in [2, ny]; j = 1, ny+1 keep set_BC's 0
        if (own && j >= jlo && j <= jhi && r >= g.own_lo && r <= g.own_hi) {
          if (urow && ur[q] * c.dt > c.cfl_x) viol++;
          if (j >= 2 && vr[q] * c.dt > c.cfl_y) viol++;
        }
        p1[q] = pr[q];
        rho1[q] = rhor[q];
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
    }
    T out[V];
    const int io = r - 3;
    if (YFIRST) {
      // y sweep of row r in front of the pipeline; rows outside [ilo, ihi] (the ghost rows) enter
      // unswept, which is what the twin buffer holds for the x sweep in the two-kernel form
      T Fp[V];
      bool rz = true;
#pragma unroll
      for (int q = 0; q < V; ++q) rz = rz && Fr[q] == (T)0;
      if (r < ilo || r > ihi || __all(rz)) {
#pragma unroll
        for (int q = 0; q < V; ++q) Fp[q] = Fr[q];
      } else {
        fct_y_row<T, V, false>(c, j0, ny, Fr, vr, Fp);
      }
      pipe.template push<true>(c, r, ilo, ihi, Fp, ur, out);
    } else {
      T Fp[V];
      pipe.template push<false>(c, r, ilo, ihi, Fr, ur, Fp);   // F'[r-3]
      if (io >= ra && io <= rb) {
        bool rz = true;
#pragma unroll
        for (int q = 0; q < V; ++q) rz = rz && Fp[q] == (T)0;
        if (__all(rz)) {
#pragma unroll
          for (int q = 0; q < V; ++q) out[q] = (T)0;
        } else {
          fct_y_row<T, V, true>(c, j0, ny, Fp, v3, out);
        }
      }
#pragma unroll
      for (int q = 0; q < V; ++q) {
        v3[q] = v2[q]; v2[q] = v1[q]; v1[q] = vr[q];
      }
    }
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
