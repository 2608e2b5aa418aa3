// issue_rates.hip -- what one VALU / SALU wave instruction costs on gfx950, by type, dependent and
// independent, at 1..4 waves per SIMD.  Every wave times its own loop with s_memtime (shader cycles),
// so the figures do not depend on the clock the chip happens to run at.
//
//   hipcc --offload-arch=gfx950 -O2 -o issue_rates issue_rates.hip && ./issue_rates
//
// Output: cycles per wave instruction as seen by ONE wave (latency-bound when dependent, issue-bound
// when independent) and the per-SIMD throughput figure cycles / (instructions of all waves on the SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

enum Op { FMA64, ADD64, MUL64, MAX64, CMP64, CND32, DPP32, ADDU32, FMA32, PKFMA32, RCP64, LDEXP64, SMUL, MIX_FMA64_SALU, MIX_ADD64_DPP, MIN3MAX, NOPS };
static const char* kNames[NOPS] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "v_max_f64", "v_cmp_lt_f64", "v_cndmask_b32", "v_mov_b32 dpp wave_shr",
                                   "v_add_u32", "v_fma_f32", "v_pk_fma_f32", "v_rcp_f64", "v_ldexp_f64", "s_mul_i32", "v_fma_f64 + s_mul_i32 (1:1)",
                                   "v_add_f64 + v_mov dpp (1:1)", "v_max_f64+v_min_f64 (1:1)"};

template <int OP, int ILP>
__global__ __launch_bounds__(256) void k(unsigned long long* out, int iters, double seed) {
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  const double b = 1.0000001, c = 1e-9;
  float f0 = (float)a0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
  typedef float f2v __attribute__((ext_vector_type(2)));
  f2v p0 = {f0, f1}, p1 = {f2, f3}, p2 = {f1, f0}, p3 = {f3, f2};
  const f2v pb = {1.0000001f, 1.0000001f}, pc = {1e-9f, 1e-9f};
  int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
  int s0 = blockIdx.x | 1, s1 = 3, s2 = 5, s3 = 7;
  unsigned long long w0 = wall_clock64();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#define CH(n) (ILP == 1 ? 0 : (n) % ILP)
#define ONE(n)                                                                                                     \
  {                                                                                                                \
    if (OP == FMA64) { if (CH(n) == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c)); else if (CH(n) == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c)); else if (CH(n) == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c)); else asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c)); } \
    if (OP == ADD64) { if (CH(n) == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a0) : "v"(c)); else if (CH(n) == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a1) : "v"(c)); else if (CH(n) == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a2) : "v"(c)); else asm volatile("v_add_f64 %0, %0, %1" : "+v"(a3) : "v"(c)); } \
    if (OP == MUL64) { if (CH(n) == 0) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a0) : "v"(b)); else if (CH(n) == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a1) : "v"(b)); else if (CH(n) == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a2) : "v"(b)); else asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a3) : "v"(b)); } \
    if (OP == MAX64) { if (CH(n) == 0) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a0) : "v"(b)); else if (CH(n) == 1) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a1) : "v"(b)); else if (CH(n) == 2) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a2) : "v"(b)); else asm volatile("v_max_f64 %0, %0, %1" : "+v"(a3) : "v"(b)); } \
    if (OP == CMP64) { asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(CH(n) == 0 ? a0 : CH(n) == 1 ? a1 : CH(n) == 2 ? a2 : a3), "v"(b) : "vcc"); } \
    if (OP == CND32) { if (CH(n) == 0) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i0) : "v"(i3) : ); else if (CH(n) == 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i1) : "v"(i3)); else if (CH(n) == 2) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i2) : "v"(i3)); else asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i3) : "v"(i0)); } \
    if (OP == DPP32) { if (CH(n) == 0) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(i0)); else if (CH(n) == 1) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(i1)); else if (CH(n) == 2) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(i2)); else asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(i3)); } \
    if (OP == ADDU32) { if (CH(n) == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(i0) : "v"(i3)); else if (CH(n) == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(i1) : "v"(i3)); else if (CH(n) == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(i2) : "v"(i3)); else asm volatile("v_add_u32 %0, %0, %1" : "+v"(i3) : "v"(i0)); } \
    if (OP == FMA32) { if (CH(n) == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(1.0000001f), "v"(1e-9f)); else if (CH(n) == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f1) : "v"(1.0000001f), "v"(1e-9f)); else if (CH(n) == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f2) : "v"(1.0000001f), "v"(1e-9f)); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f3) : "v"(1.0000001f), "v"(1e-9f)); } \
    if (OP == PKFMA32) { if (CH(n) == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(pb), "v"(pc)); else if (CH(n) == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(pb), "v"(pc)); else if (CH(n) == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(pb), "v"(pc)); else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(pb), "v"(pc)); } \
    if (OP == RCP64) { if (CH(n) == 0) asm volatile("v_rcp_f64 %0, %0" : "+v"(a0)); else if (CH(n) == 1) asm volatile("v_rcp_f64 %0, %0" : "+v"(a1)); else if (CH(n) == 2) asm volatile("v_rcp_f64 %0, %0" : "+v"(a2)); else asm volatile("v_rcp_f64 %0, %0" : "+v"(a3)); } \
    if (OP == LDEXP64) { if (CH(n) == 0) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(a0)); else if (CH(n) == 1) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(a1)); else if (CH(n) == 2) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(a2)); else asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(a3)); } \
    if (OP == SMUL) { if (CH(n) == 0) asm volatile("s_mul_i32 %0, %0, %1" : "+s"(s0) : "s"(s3)); else if (CH(n) == 1) asm volatile("s_mul_i32 %0, %0, %1" : "+s"(s1) : "s"(s3)); else if (CH(n) == 2) asm volatile("s_mul_i32 %0, %0, %1" : "+s"(s2) : "s"(s3)); else asm volatile("s_mul_i32 %0, %0, %1" : "+s"(s3) : "s"(s0)); } \
    if (OP == MIX_FMA64_SALU) { if (CH(n) == 0) asm volatile("v_fma_f64 %0, %0, %2, %3\n s_mul_i32 %1, %1, %4" : "+v"(a0), "+s"(s0) : "v"(b), "v"(c), "s"(s3)); else if (CH(n) == 1) asm volatile("v_fma_f64 %0, %0, %2, %3\n s_mul_i32 %1, %1, %4" : "+v"(a1), "+s"(s1) : "v"(b), "v"(c), "s"(s3)); else if (CH(n) == 2) asm volatile("v_fma_f64 %0, %0, %2, %3\n s_mul_i32 %1, %1, %4" : "+v"(a2), "+s"(s2) : "v"(b), "v"(c), "s"(s3)); else asm volatile("v_fma_f64 %0, %0, %2, %3\n s_mul_i32 %1, %1, %4" : "+v"(a3), "+s"(s3) : "v"(b), "v"(c), "s"(s0)); } \
    if (OP == MIX_ADD64_DPP) { if (CH(n) == 0) asm volatile("v_add_f64 %0, %0, %2\n v_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a0), "+v"(i0) : "v"(c)); else if (CH(n) == 1) asm volatile("v_add_f64 %0, %0, %2\n v_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a1), "+v"(i1) : "v"(c)); else if (CH(n) == 2) asm volatile("v_add_f64 %0, %0, %2\n v_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a2), "+v"(i2) : "v"(c)); else asm volatile("v_add_f64 %0, %0, %2\n v_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a3), "+v"(i3) : "v"(c)); } \
    if (OP == MIN3MAX) { if (CH(n) == 0) asm volatile("v_max_f64 %0, %0, %1\n v_min_f64 %0, %0, %2" : "+v"(a0) : "v"(b), "v"(c)); else if (CH(n) == 1) asm volatile("v_max_f64 %0, %0, %1\n v_min_f64 %0, %0, %2" : "+v"(a1) : "v"(b), "v"(c)); else if (CH(n) == 2) asm volatile("v_max_f64 %0, %0, %1\n v_min_f64 %0, %0, %2" : "+v"(a2) : "v"(b), "v"(c)); else asm volatile("v_max_f64 %0, %0, %1\n v_min_f64 %0, %0, %2" : "+v"(a3) : "v"(b), "v"(c)); } \
  }
    ONE(0) ONE(1) ONE(2) ONE(3) ONE(4) ONE(5) ONE(6) ONE(7) ONE(8) ONE(9) ONE(10) ONE(11) ONE(12) ONE(13) ONE(14) ONE(15)
    ONE(16) ONE(17) ONE(18) ONE(19) ONE(20) ONE(21) ONE(22) ONE(23) ONE(24) ONE(25) ONE(26) ONE(27) ONE(28) ONE(29) ONE(30) ONE(31)
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  unsigned long long w1 = wall_clock64();
  // keep everything alive
  double sink = a0 + a1 + a2 + a3 + f0 + f1 + f2 + f3 + p0.x + p1.x + p2.y + p3.y + i0 + i1 + i2 + i3 + s0 + s1 + s2 + s3;
  if (sink == 123.456) out[1] = 1;
  if ((threadIdx.x & 63) == 0) { out[1 + blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0; if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = w1 - w0; }
}

template <int OP, int ILP>
void run(unsigned long long* d, int wps) {
  const int iters = 2000, per_iter = 32;
  const int blocks = 256 * wps;   // 256 CUs x wps blocks of 4 waves = wps waves per SIMD
  hipMemset(d, 0, 8 * (1 + blocks * 4));
  hipLaunchKernelGGL((k<OP, ILP>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(1 + blocks * 4);
  hipMemcpy(h.data(), d, 8 * h.size(), hipMemcpyDeviceToHost);
  std::vector<double> cyc;
  for (size_t i = 1; i < h.size(); ++i) cyc.push_back((double)h[i]);
  std::sort(cyc.begin(), cyc.end());
  const double med = cyc[cyc.size() / 2];
  const int mult = (OP == MIX_FMA64_SALU || OP == MIX_ADD64_DPP || OP == MIN3MAX) ? 2 : 1;
  const double n = (double)iters * per_iter * mult;
  printf("%-30s ILP %d  waves/SIMD %d : %6.2f cycles per instruction per wave, %6.2f per SIMD issue   (shader clock %.2f GHz)\n", kNames[OP], ILP, wps,
         med / n, med / n / wps, (double)h[1] / ((double)h[0] * 10.0));
}

template <int OP>
void both(unsigned long long* d) {
  for (int w : {1, 2, 3, 4}) run<OP, 1>(d, w);
  for (int w : {1, 2, 3, 4}) run<OP, 4>(d, w);
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, 8 * (1 + 256 * 8 * 4));
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s, %d CUs, clock %d kHz; s_memtime counts at a constant 100 MHz on gfx9 if the figures below look 24x too small\n", p.gcnArchName,
         p.multiProcessorCount, p.clockRate);
  both<FMA64>(d); both<ADD64>(d); both<MUL64>(d); both<MAX64>(d); both<CMP64>(d); both<CND32>(d); both<DPP32>(d); both<ADDU32>(d);
  both<FMA32>(d); both<PKFMA32>(d); both<RCP64>(d); both<LDEXP64>(d); both<SMUL>(d); both<MIX_FMA64_SALU>(d); both<MIX_ADD64_DPP>(d);
  both<MIN3MAX>(d);
  return 0;
}
