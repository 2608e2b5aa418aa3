// runtime/selftest.h -- device side of vof_selftest_division
//
// Part of the host-side runtime of libvof2d_hip.so; included (once, in this order) by vof2d_api.hip:
// context.h, launches.h, schedule.h, comm.h, selftest.h.  Everything here has internal linkage.
#pragma once
#include "context.h"

namespace {

// ---- self-test of the exact constant-denominator division (vof2d_kernels.h div_by_const) against
// the hardware IEEE division, on adversarial numerators: subnormal quotients at and next to the
// midpoints of the subnormal grid (the double-rounding case), tiny / huge / special values.
__device__ inline uint64_t mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
template <typename T> struct SelfT;
template <> struct SelfT<double> { static constexpr int emin = -1074, kbits = 51, ebig = 1000; };
template <> struct SelfT<float> { static constexpr int emin = -149, kbits = 22, ebig = 120; };
template <typename T>
__global__ void k_selftest_division(uint64_t seed, int64_t n, T* __restrict__ oa, T* __restrict__ ob, T* __restrict__ oq) {
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= n) return;
  using S = SelfT<T>;
  uint64_t h1 = mix64(seed + 4 * (uint64_t)id), h2 = mix64(h1), h3 = mix64(h2), h4 = mix64(h3);
  auto unit = [](uint64_t h) { return (T)(1.0 + (double)(h >> 11) * 0x1p-53); };  // [1, 2), full significand
  const int cat = (int)(id % 8);
  // denominators: |b| in [1, 2^37) like ap of the Jacobi stencil, or in (2^-40, 1) like dx, dt, dx*dy
  const bool small_b = (cat == 3) || (cat == 7) || (cat == 0 && (h4 & 1));
  T b = unit(h1) * (T)__builtin_ldexp(1.0, small_b ? -1 - (int)(h2 % 40) : (int)(h2 % 37));
  if (h2 & (1ull << 50)) b = -b;
  T a;
  if (cat == 0 || cat == 7) {                       // ordinary magnitudes over the whole range
    a = unit(h3) * (T)__builtin_ldexp(1.0, (int)(h4 % (2 * S::ebig)) - S::ebig);
  } else if (cat == 1) {                            // tiny numerators down to the smallest subnormal
    a = unit(h3) * (T)__builtin_ldexp(1.0, S::emin + (int)(h4 % 200));
  } else if (cat == 2 || cat == 5 || cat == 6) {    // subnormal quotient next to / on a grid midpoint
    if (cat == 5) b = (T)(double)(1 + (h1 % 4095)) * (T)__builtin_ldexp(1.0, (int)(h2 % 20));  // exact ties
    const int kb = 1 + (int)(h4 % S::kbits);
    const double k = (double)(h3 >> (64 - kb)) + 0.5;       // midpoint index + 1/2
    double nn = __builtin_rint(k * (double)dabs<T>(b));      // numerator in units of the smallest subnormal
    if (cat == 6) nn += (double)((int)(h4 >> 60) - 8);       // a few units beside it
    a = (T)__builtin_ldexp(nn, S::emin);
  } else if (cat == 3) {                            // huge numerators over small denominators
    a = unit(h3) * (T)__builtin_ldexp(1.0, S::ebig - (int)(h4 % 100) + (sizeof(T) == 8 ? 23 : 7));
  } else {                                          // zeros, infinities, NaN
    const T sp[6] = {(T)0.0, (T)-0.0, DivLimits<T>::inf, -DivLimits<T>::inf, (T)__builtin_nan(""), DivLimits<T>::denorm_min};
    a = sp[h3 % 6];
  }
  if (h3 & (1ull << 40)) a = -a;
  const T y = (T)1.0 / b;
  oa[id] = a;
  ob[id] = b;
  // |b| < 1: the scalar routine with the huge-numerator tier.  |b| >= 1: the V-wide form the fused
  // Jacobi kernel uses (wave-level branches), fed with this lane's and its neighbour's operands --
  // the categories alternate by lane, so tiny, ordinary, special and tie numerators meet in one wave.
  T q = div_by_const<T, true>(a, b, y);
  {
    const T a2 = __shfl_xor(a, 1, 64), b2 = __shfl_xor(b, 1, 64), y2 = __shfl_xor(y, 1, 64);
    const T av[2] = {a, a2}, bv[2] = {b, b2}, yv[2] = {y, y2};
    T rv[2];
    div_by_const_v<T, 2, false>(rv, av, bv, yv);
    if (!(dabs<T>(b) < (T)1)) q = ((id >> 3) & 1) ? rv[0] : div_by_const<T, false>(a, b, y);   // both forms get checked
  }
  oq[id] = q;
}

}  // namespace
