// Probe: semantics of __builtin_amdgcn_global_load_lds on gfx950 (used to design the LDS-staged
// prefetch of the marching kernels).  Each wave DMAs 1 KiB (64 lanes x 16 B) + a 16-B edge block
// (4 lanes x 4 B) into its own LDS slot, waits on vmcnt, and reads it back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void probe(const double* __restrict__ src, double* __restrict__ out, int ncols) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  char* slot = smem + wave * 1040;  // [16 B edge][1024 B main]
  const double* row = src + 16 + wave * 128;  // pretend tile start
  // main: lane l -> 16 bytes at slot + 16 + l*16
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(row + 2 * lane),
                                   (__attribute__((address_space(3))) void*)(slot + 16), 16, 0, 0);
  // edge: lanes 0,1 -> left neighbour (row[-1]) as two dwords; lanes 2,3 -> right neighbour (row[128])
  if (lane < 4) {
    const float* g = (lane < 2) ? (const float*)(row - 1) + lane : (const float*)(row + 128) + (lane - 2);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)slot, 4, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const double2 c = *reinterpret_cast<const double2*>(slot + 16 + lane * 16);
  const double l = *reinterpret_cast<const double*>(lane == 0 ? slot : slot + 16 + lane * 16 - 8);
  const double r = *reinterpret_cast<const double*>(lane == 63 ? slot + 8 : slot + 16 + lane * 16 + 16);
  double* o = out + (size_t)(wave * 64 + lane) * 4;
  o[0] = l; o[1] = c.x; o[2] = c.y; o[3] = r;
}

int main() {
  const int n = 16 + 4 * 128 + 16;
  std::vector<double> h(n);
  for (int i = 0; i < n; ++i) h[i] = i;
  double *d, *o;
  hipMalloc(&d, n * 8); hipMalloc(&o, 256 * 4 * 8);
  hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(256), 4 * 1040, 0, d, o, 128);
  std::vector<double> r(256 * 4);
  hipError_t e = hipMemcpy(r.data(), o, 256 * 4 * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < 4; ++w)
    for (int l = 0; l < 64; ++l) {
      const double* q = &r[(w * 64 + l) * 4];
      double base = 16 + w * 128 + 2 * l;
      if (q[0] != base - 1 || q[1] != base || q[2] != base + 1 || q[3] != base + 2) {
        if (bad < 8) printf("wave %d lane %d: got %g %g %g %g expected %g..%g\n", w, l, q[0], q[1], q[2], q[3], base - 1, base + 2);
        ++bad;
      }
    }
  printf("lds_dma_probe: %s (%d mismatches, hip status %d)\n", bad ? "FAIL" : "OK", bad, (int)e);
  return bad != 0;
}
