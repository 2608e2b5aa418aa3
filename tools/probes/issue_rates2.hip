// issue_rates2.hip -- second pass of issue_rates.hip with the whole loop body as ONE asm block (the compiler puts an
// s_nop between separate asm statements), for the instruction patterns of the VOF kernels: selects, compares, DPP.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

// BODY uses v[10:25] as scratch (declared clobbered), s[20:27], vcc
#define KERNEL(NAME, N_PER_REPT, BODY)                                                                        \
  __global__ __launch_bounds__(256) void NAME(unsigned long long* out, int iters) {                            \
    unsigned long long w0 = wall_clock64();                                                                    \
    unsigned long long t0 = __builtin_readcyclecounter();                                                      \
    for (int it = 0; it < iters; ++it) {                                                                       \
      asm volatile(".rept 8\n" BODY ".endr\n" ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", \
                   "v22", "v23", "v24", "v25", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "vcc");      \
    }                                                                                                          \
    unsigned long long t1 = __builtin_readcyclecounter();                                                      \
    unsigned long long w1 = wall_clock64();                                                                    \
    if ((threadIdx.x & 63) == 0) {                                                                             \
      out[1 + blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                                                  \
      if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = w1 - w0;                                               \
    }                                                                                                          \
  }                                                                                                            \
  static const int NAME##_n = N_PER_REPT * 8;

KERNEL(k_fma64_dep, 4, "v_fma_f64 v[10:11], v[10:11], v[12:13], v[14:15]\n v_fma_f64 v[10:11], v[10:11], v[12:13], v[14:15]\n v_fma_f64 v[10:11], v[10:11], v[12:13], v[14:15]\n v_fma_f64 v[10:11], v[10:11], v[12:13], v[14:15]\n")
KERNEL(k_fma64_ilp4, 4, "v_fma_f64 v[10:11], v[10:11], v[12:13], v[14:15]\n v_fma_f64 v[16:17], v[16:17], v[12:13], v[14:15]\n v_fma_f64 v[18:19], v[18:19], v[12:13], v[14:15]\n v_fma_f64 v[20:21], v[20:21], v[12:13], v[14:15]\n")
KERNEL(k_add64_ilp2, 4, "v_add_f64 v[10:11], v[10:11], v[12:13]\n v_add_f64 v[16:17], v[16:17], v[12:13]\n v_add_f64 v[10:11], v[10:11], v[12:13]\n v_add_f64 v[16:17], v[16:17], v[12:13]\n")
KERNEL(k_cnd_vcc_dep, 4, "v_cndmask_b32 v10, v10, v12, vcc\n v_cndmask_b32 v10, v10, v12, vcc\n v_cndmask_b32 v10, v10, v12, vcc\n v_cndmask_b32 v10, v10, v12, vcc\n")
KERNEL(k_cnd_vcc_ilp4, 4, "v_cndmask_b32 v10, v10, v12, vcc\n v_cndmask_b32 v11, v11, v12, vcc\n v_cndmask_b32 v13, v13, v12, vcc\n v_cndmask_b32 v14, v14, v12, vcc\n")
KERNEL(k_cnd_sgpr_ilp4, 4, "v_cndmask_b32_e64 v10, v10, v12, s[20:21]\n v_cndmask_b32_e64 v11, v11, v12, s[20:21]\n v_cndmask_b32_e64 v13, v13, v12, s[20:21]\n v_cndmask_b32_e64 v14, v14, v12, s[20:21]\n")
KERNEL(k_cmp_cnd2, 6, "v_cmp_lt_f64 vcc, v[10:11], v[12:13]\n v_cndmask_b32 v14, v14, v16, vcc\n v_cndmask_b32 v15, v15, v17, vcc\n v_cmp_lt_f64 s[20:21], v[18:19], v[12:13]\n v_cndmask_b32_e64 v20, v20, v16, s[20:21]\n v_cndmask_b32_e64 v21, v21, v17, s[20:21]\n")
KERNEL(k_cmp64_vcc, 4, "v_cmp_lt_f64 vcc, v[10:11], v[12:13]\n v_cmp_lt_f64 vcc, v[14:15], v[12:13]\n v_cmp_lt_f64 vcc, v[16:17], v[12:13]\n v_cmp_lt_f64 vcc, v[18:19], v[12:13]\n")
KERNEL(k_cmp64_sgpr, 4, "v_cmp_lt_f64 s[20:21], v[10:11], v[12:13]\n v_cmp_lt_f64 s[22:23], v[14:15], v[12:13]\n v_cmp_lt_f64 s[24:25], v[16:17], v[12:13]\n v_cmp_lt_f64 s[26:27], v[18:19], v[12:13]\n")
KERNEL(k_cmp32_vcc, 4, "v_cmp_lt_f32 vcc, v10, v12\n v_cmp_lt_f32 vcc, v14, v12\n v_cmp_lt_f32 vcc, v16, v12\n v_cmp_lt_f32 vcc, v18, v12\n")
KERNEL(k_mov_ilp4, 4, "v_mov_b32 v10, v12\n v_mov_b32 v11, v13\n v_mov_b32 v14, v15\n v_mov_b32 v16, v17\n")
KERNEL(k_dpp_ilp4, 4, "v_mov_b32_dpp v10, v12 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v11, v13 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v14, v15 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v16, v17 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
KERNEL(k_dpp_rowshr_ilp4, 4, "v_mov_b32_dpp v10, v12 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v11, v13 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v14, v15 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v16, v17 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
KERNEL(k_bfi_ilp4, 4, "v_bfi_b32 v10, v12, v13, v10\n v_bfi_b32 v11, v12, v13, v11\n v_bfi_b32 v14, v12, v13, v14\n v_bfi_b32 v16, v12, v13, v16\n")
KERNEL(k_and_ilp4, 4, "v_and_b32 v10, v12, v10\n v_and_b32 v11, v12, v11\n v_and_b32 v14, v12, v14\n v_and_b32 v16, v12, v16\n")
KERNEL(k_max64_ilp4, 4, "v_max_f64 v[10:11], v[10:11], v[12:13]\n v_max_f64 v[16:17], v[16:17], v[12:13]\n v_max_f64 v[18:19], v[18:19], v[12:13]\n v_max_f64 v[20:21], v[20:21], v[12:13]\n")
KERNEL(k_mul64_sgpr_ilp4, 4, "v_mul_f64 v[10:11], s[20:21], v[10:11]\n v_mul_f64 v[16:17], s[20:21], v[16:17]\n v_mul_f64 v[18:19], s[22:23], v[18:19]\n v_mul_f64 v[20:21], s[22:23], v[20:21]\n")
KERNEL(k_salu_ilp4, 4, "s_add_i32 s20, s20, s24\n s_add_i32 s21, s21, s24\n s_add_i32 s22, s22, s24\n s_add_i32 s23, s23, s24\n")
KERNEL(k_salu_cselect, 4, "s_cmp_lt_i32 s20, s24\n s_cselect_b32 s21, s22, s23\n s_cmp_lt_i32 s21, s24\n s_cselect_b32 s20, s22, s23\n")
KERNEL(k_valu_salu_1to1, 8, "v_fma_f64 v[10:11], v[10:11], v[12:13], v[14:15]\n s_add_i32 s20, s20, s24\n v_fma_f64 v[16:17], v[16:17], v[12:13], v[14:15]\n s_add_i32 s21, s21, s24\n v_fma_f64 v[18:19], v[18:19], v[12:13], v[14:15]\n s_add_i32 s22, s22, s24\n v_fma_f64 v[20:21], v[20:21], v[12:13], v[14:15]\n s_add_i32 s23, s23, s24\n")
KERNEL(k_fma32_ilp4, 4, "v_fma_f32 v10, v10, v12, v14\n v_fma_f32 v16, v16, v12, v14\n v_fma_f32 v18, v18, v12, v14\n v_fma_f32 v20, v20, v12, v14\n")
KERNEL(k_cnd_after_valu, 4, "v_add_u32 v16, v16, v12\n v_cndmask_b32 v10, v10, v12, vcc\n v_add_u32 v17, v17, v12\n v_cndmask_b32 v11, v11, v12, vcc\n")
KERNEL(k_readlane, 4, "v_readlane_b32 s20, v10, 3\n v_readlane_b32 s21, v11, 3\n v_readlane_b32 s22, v12, 3\n v_readlane_b32 s23, v13, 3\n")
KERNEL(k_readfirstlane, 4, "v_readfirstlane_b32 s20, v10\n v_readfirstlane_b32 s21, v11\n v_readfirstlane_b32 s22, v12\n v_readfirstlane_b32 s23, v13\n")
KERNEL(k_saveexec, 4, "s_and_saveexec_b64 s[20:21], vcc\n s_mov_b64 exec, s[20:21]\n s_and_saveexec_b64 s[22:23], vcc\n s_mov_b64 exec, s[22:23]\n")
KERNEL(k_div_scale, 4, "v_div_scale_f64 v[10:11], vcc, v[12:13], v[14:15], v[12:13]\n v_div_scale_f64 v[16:17], vcc, v[12:13], v[14:15], v[12:13]\n v_div_fmas_f64 v[18:19], v[18:19], v[12:13], v[14:15]\n v_div_fixup_f64 v[20:21], v[20:21], v[12:13], v[14:15]\n")

template <typename K>
void run(const char* name, K kern, int n_per_iter, unsigned long long* d, int wps) {
  const int iters = 2000;
  const int blocks = 256 * wps;
  hipMemset(d, 0, 8 * (1 + blocks * 4));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(1 + blocks * 4);
  hipMemcpy(h.data(), d, 8 * h.size(), hipMemcpyDeviceToHost);
  std::vector<double> cyc;
  double tot = 0;
  for (size_t i = 1; i < h.size(); ++i) { cyc.push_back((double)h[i]); tot += (double)h[i]; }
  std::sort(cyc.begin(), cyc.end());
  const double med = cyc[cyc.size() / 2];
  const double n = (double)iters * n_per_iter;
  printf("%-20s waves/SIMD %d : %6.2f cycles per instruction per wave (median; min %.2f max %.2f), %6.2f per SIMD\n", name, wps, med / n,
         cyc.front() / n, cyc.back() / n, med / n / wps);
}
#define RUN(NAME) for (int w : {1, 2, 3, 4}) run(#NAME, NAME, NAME##_n, d, w);

int main() {
  unsigned long long* d;
  hipMalloc(&d, 8 * (1 + 256 * 8 * 4));
  RUN(k_fma64_dep) RUN(k_fma64_ilp4) RUN(k_add64_ilp2) RUN(k_cnd_vcc_dep) RUN(k_cnd_vcc_ilp4) RUN(k_cnd_sgpr_ilp4) RUN(k_cmp_cnd2)
  RUN(k_cmp64_vcc) RUN(k_cmp64_sgpr) RUN(k_cmp32_vcc) RUN(k_mov_ilp4) RUN(k_dpp_ilp4) RUN(k_dpp_rowshr_ilp4) RUN(k_bfi_ilp4) RUN(k_and_ilp4)
  RUN(k_max64_ilp4) RUN(k_mul64_sgpr_ilp4) RUN(k_salu_ilp4) RUN(k_salu_cselect) RUN(k_valu_salu_1to1) RUN(k_fma32_ilp4) RUN(k_cnd_after_valu)
  RUN(k_readlane) RUN(k_readfirstlane) RUN(k_saveexec) RUN(k_div_scale)
  return 0;
}
