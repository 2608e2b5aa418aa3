// vof2d_device.h -- shared host/device definitions for the gfx950 kernels.
//
// Data layout in HBM (DESIGN.md "layout"): every field is a pitched 2-D array
//   element (i, j)  ->  base[(i - row_lo) * pitch + col0 + j]
// with j (y) contiguous, exactly the reference's (nx+2, ny+2) row-major order
// (2dvof.py:53) plus padding: col0 makes interior column j = 1
// start on a 128-byte boundary and the pitch is a multiple of 128 bytes, so a
// wave's 16-byte-per-lane accesses are aligned and never straddle rows.
//
// Work decomposition: one 64-lane wave owns a tile of 64*V contiguous columns
// (V = 2 elements per lane, VecWidth below) and marches along
// i over a chunk of rows, keeping the i-1 / i / i+1 rows in registers so each
// element is fetched from HBM once.  j-1 / j+V neighbours come from L1 hits or
// cross-lane shuffles.  Waves never synchronise with each other (no LDS, no
// barriers), so a 256-thread block is just four independent adjacent tiles.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vof {

struct Geom {
  int nx, ny;            // global interior cells
  int row_lo, row_hi;    // stored rows (global, inclusive)
  int ilo, ihi;          // computable interior rows: max(1,row_lo+1) .. min(nx,row_hi-1)
  int own_lo, own_hi;    // rows whose counters/residuals this handle reports
  int wall_lo, wall_hi;  // 1 if this strip holds the i = 0 / i = nx+1 ghost row
  int64_t pitch;         // elements per stored row
  int col0;              // column offset of j = 0
  int ntj;               // column tiles of 64*V columns covering j = 1 .. ny
};

// Constants rounded once from Python-double folded values.
template <typename T>
struct Consts {
  T dt, dx, dy, dxi, dyi, dxi2, dyi2, rho_l, rho_g, nu_l, nu_g, sigma, gx, gy;
  T nrm_x, nrm_y, kap_x, kap_y, dxdy, dtdy, dtdx, cfl_x, cfl_y, half_dx, half_dy, sqrt2dx, tiny;
  T inv_dx, inv_dy, inv_dt, inv_dxdy;  // correctly rounded reciprocals (div_by_const)
  T dt_rho_l, dt_rho_g;                // dt / rho_l, dt / rho_g: update_uv's dt / r (2dvof.py:273) where both cells hold pure liquid / gas
  // set_init_F literals (2dvof.py:141-159), folded in double then rounded
  T ic1_x2, ic1_y2, ic_r, ic_cx, ic2_cy, ic3_cy, ic3_pool;
};

// Elements per lane and row: 2 for both precisions, i.e. a wave tile is 128 columns.  (16 bytes per
// lane -- 4 floats, 256-column tiles -- halves the number of tiles, and with it the chunk length a
// residency round allows, so the lead-in rows of every chunk weigh twice as much: fp32 at 4096^2
// 601 us/step with V = 4, 443 us/step with V = 2; 2048^2 249 -> 174 us.)
template <typename T> struct VecWidth { static constexpr int V = 2; };

// Columns on each side of a 64*V-column tile that k_momentum / the FCT y stage load and compute but do not store
// (the neighbouring tile does).  The stencils need 2 and 4; 8 makes the tile stride 112 columns = 7 cache lines of
// doubles, so every tile's stored segment starts and ends on a 128-byte line (on a 64-byte sector in fp32).  With
// strides of 124 / 120 columns two waves -- often on different CUs -- each write part of the line between their tiles:
// the compute-free skeletons of the two kernels (tools/probes/stream_pattern.hip) run 11 % / 7 % faster at 112.
struct TileHalo { static constexpr int momentum = 2, transport = 8; };

template <typename T, int V>
struct alignas(sizeof(T) * V) Pack {
  T v[V];
};

// ti.max / ti.min.  The oracle evaluates them as comparisons (a > b ? a : b); v_max_f64 / v_min_f64
// return the same VALUE for every non-NaN pair (only the sign of a +-0 tie can differ, which no
// later operation observes) at a third of the instructions of compare + 2x v_cndmask_b32.
template <typename T> __device__ __forceinline__ T vmax(T a, T b);
template <typename T> __device__ __forceinline__ T vmin(T a, T b);
template <> __device__ __forceinline__ double vmax<double>(double a, double b) { return __builtin_fmax(a, b); }
template <> __device__ __forceinline__ double vmin<double>(double a, double b) { return __builtin_fmin(a, b); }
template <> __device__ __forceinline__ float vmax<float>(float a, float b) { return __builtin_fmaxf(a, b); }
template <> __device__ __forceinline__ float vmin<float>(float a, float b) { return __builtin_fminf(a, b); }

// 2dvof.py:192-195  var(a, b, c) = a + b + c - max(a,b,c) - min(a,b,c), left to right
template <typename T> __device__ __forceinline__ T var3(T a, T b, T c) {
  return ((a + b) + c) - vmax(vmax(a, b), c) - vmin(vmin(a, b), c);
}

// 2dvof.py:201-203 for one cell: rho, nu from F
template <typename T> __device__ __forceinline__ T rho_of(const Consts<T>& c, T F) {
  T Fc = var3((T)0.0, (T)1.0, F);
  return c.rho_g * ((T)1 - Fc) + c.rho_l * Fc;
}
template <typename T> __device__ __forceinline__ T nu_of(const Consts<T>& c, T F) {
  T Fc = var3((T)0.0, (T)1.0, F);
  return c.nu_l * Fc + c.nu_g * ((T)1.0 - Fc);
}

template <typename T> __device__ __forceinline__ T dsqrt(T x);
template <> __device__ __forceinline__ double dsqrt<double>(double x) { return __builtin_sqrt(x); }
template <> __device__ __forceinline__ float dsqrt<float>(float x) { return __builtin_sqrtf(x); }
template <typename T> __device__ __forceinline__ T dabs(T x);
template <> __device__ __forceinline__ double dabs<double>(double x) { return __builtin_fabs(x); }
template <> __device__ __forceinline__ float dabs<float>(float x) { return __builtin_fabsf(x); }

}  // namespace vof
