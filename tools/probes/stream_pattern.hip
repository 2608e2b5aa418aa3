// stream_pattern.hip -- what the memory system gives a marching kernel of the VOF step's SHAPE when the arithmetic is
// taken away: NIN input arrays and NOUT output arrays of nx x ny doubles, one wave = 128 columns (16 B per lane)
// marching along i over a chunk of R rows after L lead-in rows (re-read, nothing stored), next row prefetched one
// (or D) iterations ahead, nontemporal stores, blocks of 4 adjacent tiles in index order -- the structure of
// k_momentum (3 in / 3 out, L = 7), k_transport (4 / 3, L = 6), k_jacobi_tb (2 / 1, L = 10).
//
//   hipcc --offload-arch=gfx950 -O3 -o stream_pattern stream_pattern.hip
//   ./stream_pattern            (prints one line per configuration: us, TB/s on algorithmic and on requested bytes)
//
// Knobs per run: R, L, waves per SIMD (capped with dynamic LDS), persistent waves pulling chunks from a queue,
// prefetch depth, nontemporal loads on/off.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

typedef double v2d __attribute__((ext_vector_type(2)));

struct Args {
  const double* in[4];
  double* out[4];
  long pitch;
  int nx, ny, ntiles, R, L, nchunks;
  unsigned int* queue;   // [0] next item, [1] waves done (persistent mode)
  int persistent, nt_loads, guided, sync_rows, wpb, work;   // work: dependent fp64 FMAs per row and chain (2 chains per lane)
  double km, ka;
  int tstride, vlo, vhi;   // columns from one tile to the next (128: disjoint tiles), lanes that store (others only load: the tile overlap)
  int stagger;   // s_sleep units (64 cycles each) per residency slot: the waves sharing a SIMD start out of phase
};

template <int NIN, int NOUT, int D>
__device__ __forceinline__ void chunk(const Args& a, int tj, int ra, int rb, int lane) {
  const long col = 8 + (long)tj * a.tstride + lane * 2;   // 64-byte aligned start of tile 0, 16 B per lane
  const bool st_lane = lane >= a.vlo && lane <= a.vhi;
  v2d q[D][NIN];
  const int r0 = ra - a.L;
  auto ld = [&](int k, int r) -> v2d {
    const int rc = r < 0 ? 0 : (r >= a.nx ? a.nx - 1 : r);
    const v2d* p = reinterpret_cast<const v2d*>(a.in[k] + (long)rc * a.pitch + col);
    return a.nt_loads ? __builtin_nontemporal_load(p) : *p;
  };
#pragma unroll
  for (int d = 0; d < D; ++d)
#pragma unroll
    for (int k = 0; k < NIN; ++k) q[d][k] = ld(k, r0 + d);
  v2d carry = {0.0, 0.0};
  for (int r = r0; r <= rb; r += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int rr = r + d;
      if (rr > rb) break;
      v2d cur[NIN];
#pragma unroll
      for (int k = 0; k < NIN; ++k) cur[k] = q[d][k];
      if (rr + D <= rb) {
#pragma unroll
        for (int k = 0; k < NIN; ++k) q[d][k] = ld(k, rr + D);
      }
      if (a.sync_rows) __builtin_amdgcn_s_barrier();   // keep the waves of a block on the same row
      v2d s = carry;
#pragma unroll
      for (int k = 0; k < NIN; ++k) s += cur[k];
      for (int w = 0; w < a.work; ++w) {   // the arithmetic of a stencil row: two dependent chains per lane
        s.x = __builtin_fma(s.x, a.km, a.ka);
        s.y = __builtin_fma(s.y, a.km, a.ka);
      }
      carry = s * 0.5;
      if (rr >= ra) {
#pragma unroll
        for (int k = 0; k < NOUT; ++k)
          if (st_lane) __builtin_nontemporal_store(s + (double)k, reinterpret_cast<v2d*>(a.out[k] + (long)rr * a.pitch + col));
      }
    }
  }
}

template <int NIN, int NOUT, int D>
__global__ __launch_bounds__(1024) void k_stream(Args a) {
  extern __shared__ char lds_cap[];
  (void)lds_cap;
  const int lane = threadIdx.x & 63;
  if (a.stagger) {
    const int slot = (blockIdx.x >> 8) % 3;     // blocks b, b + 256, b + 512 share a CU (block b -> XCD b % 8, 32 CUs each)
    for (int k = 0; k < slot * a.stagger; ++k) __builtin_amdgcn_s_sleep(1);
  }
  if (!a.persistent) {
    const int wave = blockIdx.x * a.wpb + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tj = wave % a.ntiles, ch = wave / a.ntiles;
    // (with sync_rows every wave of the block must run the same number of iterations: nx is a multiple of R and the grid has no padding waves)
    if (ch >= a.nchunks) return;
    const int ra = ch * a.R, rb = min(ra + a.R - 1, a.nx - 1);
    chunk<NIN, NOUT, D>(a, tj, ra, rb, lane);
    return;
  }
  const unsigned total = (unsigned)a.nchunks * (unsigned)a.ntiles;
  if (a.persistent == 2) {   // static stride: wave w takes items w, w + W, w + 2 W, ... (no queue)
    const unsigned nw = gridDim.x * a.wpb;
    for (unsigned item = blockIdx.x * a.wpb + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); item < total; item += nw) {
      const int tj = item % a.ntiles, ch = item / a.ntiles;
      const int ra = ch * a.R, rb = min(ra + a.R - 1, a.nx - 1);
      chunk<NIN, NOUT, D>(a, tj, ra, rb, lane);
    }
    return;
  }
  for (;;) {
    unsigned item = 0;
    if (lane == 0) item = atomicAdd(a.queue, 1u);
    item = __builtin_amdgcn_readfirstlane(item);
    if (item >= total) break;
    const int tj = item % a.ntiles, ch = item / a.ntiles;
    const int ra = ch * a.R, rb = min(ra + a.R - 1, a.nx - 1);
    chunk<NIN, NOUT, D>(a, tj, ra, rb, lane);
  }
  if (lane == 0) {
    const unsigned nw = gridDim.x * 4;
    if (atomicAdd(a.queue + 1, 1u) == nw - 1) { a.queue[0] = 0; a.queue[1] = 0; }
  }
}

struct Cfg { int nin, nout, R, L, wps, persistent, D, nt, wpb = 4, sync = 0, nxo = 0, work = 0, stagger = 0, tstride = 128, vlo = 0, vhi = 63; };

int main(int argc, char** argv) {
  const int nx = argc > 1 ? atoi(argv[1]) : 4096, ny = nx;
  const long pitch = ny + 160;
  const size_t bytes = (size_t)(nx + 2) * pitch * 8;
  Args a{};
  std::vector<void*> bufs;
  for (int k = 0; k < 8; ++k) {
    void* p;
    hipMalloc(&p, bytes + bytes / 4);
    hipMemset(p, 0, bytes);
    bufs.push_back(p);
  }
  for (int k = 0; k < 4; ++k) a.in[k] = (const double*)bufs[k];
  for (int k = 0; k < 4; ++k) a.out[k] = (double*)bufs[4 + k];
  hipMalloc(&a.queue, 64);
  hipMemset(a.queue, 0, 64);
  a.pitch = pitch; a.nx = nx; a.ny = ny; a.ntiles = ny / 128;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<Cfg> cfgs;
  const bool fuse = argc > 2 && !strcmp(argv[2], "fuse");   // what a fused k_transport + next k_momentum would move: 4 in / 4 out against 4 / 3 + 3 / 3
  if (fuse) {
    for (int rep = 0; rep < 2; ++rep) {
      cfgs.push_back({4, 3, 16, 6, 3, 0, 1, 0, 4, 0, 0, 0, 0, 112, 4, 59});
      cfgs.push_back({3, 3, 14, 7, 3, 0, 1, 0, 4, 0, 0, 0, 0, 124, 1, 62});
      cfgs.push_back({4, 4, 16, 13, 3, 0, 1, 0, 4, 0, 0, 0, 0, 112, 4, 59});
      cfgs.push_back({4, 4, 32, 13, 3, 0, 1, 0, 4, 0, 0, 0, 0, 112, 4, 59});
      cfgs.push_back({4, 4, 32, 13, 2, 0, 1, 0, 4, 0, 0, 0, 0, 112, 4, 59});
      cfgs.push_back({4, 1, 16, 6, 3, 0, 1, 0, 4, 0, 0, 0, 0, 112, 4, 59});
    }
  }
  const bool stat = argc > 2 && !strcmp(argv[2], "static");   // persistent waves with a static stride against one wave per chunk
  if (stat) {
    for (int rep = 0; rep < 2; ++rep)
      for (int R : {16, 14, 8}) {
        const int L = 6;
        cfgs.push_back({4, 3, R, L, 3, 0, 1, 0, 4, 0, 0, 0, 0, 128, 0, 63});
        cfgs.push_back({4, 3, R, L, 3, 2, 1, 0, 4, 0, 0, 0, 0, 128, 0, 63});
        cfgs.push_back({4, 3, R, L, 3, 1, 1, 0, 4, 0, 0, 0, 0, 128, 0, 63});
        cfgs.push_back({3, 3, R, 7, 3, 0, 1, 0, 4, 0, 0, 0, 0, 128, 0, 63});
        cfgs.push_back({3, 3, R, 7, 3, 2, 1, 0, 4, 0, 0, 0, 0, 128, 0, 63});
        cfgs.push_back({4, 3, R, L, 3, 0, 1, 0, 4, 0, 0, 60, 0, 128, 0, 63});
        cfgs.push_back({4, 3, R, L, 3, 2, 1, 0, 4, 0, 0, 60, 0, 128, 0, 63});
      }
  }
  const bool mall = argc > 2 && !strcmp(argv[2], "mall");   // the same shapes on bands of rows small enough to stay in the 256 MB MALL between launches
  if (mall) {
    for (int nxo : {4096, 1024, 512, 256})
      for (int R : {2, 4, 16}) {
        cfgs.push_back({4, 3, R, R == 16 ? 6 : 0, 3, 0, 1, 0, 4, 0, nxo, 0, 0, 128, 0, 63});
        cfgs.push_back({2, 1, R, R == 16 ? 10 : 0, 3, 0, 1, 0, 4, 0, nxo, 0, 0, 128, 0, 63});
      }
  } else if (!stat && !fuse)
  for (int R : {2, 51})
    for (int L : {0, 10}) {
      if (R == 2 && L) continue;
      cfgs.push_back({2, 1, R, L, 3, 0, 1, 0, 4, 0, 0, 0, 0, 128, 0, 63});    // disjoint aligned tiles
      cfgs.push_back({2, 1, R, L, 3, 0, 1, 0, 4, 0, 0, 0, 0, 116, 3, 60});    // k_jacobi_tb: 116 of 128 columns stored
      cfgs.push_back({2, 1, R, L, 3, 0, 1, 0, 4, 0, 0, 0, 0, 120, 2, 61});    // k_transport: 120
      cfgs.push_back({2, 1, R, L, 3, 0, 1, 0, 4, 0, 0, 0, 0, 112, 4, 59});    // 112 = 7 x 128 B
      cfgs.push_back({2, 1, R, L, 3, 0, 1, 0, 4, 0, 0, 0, 0, 96, 8, 55});     // 96 = 6 x 128 B
    }
  if (!mall && !stat && !fuse) for (int ts : {128, 120, 112}) cfgs.push_back({4, 3, 16, 6, 3, 0, 1, 0, 4, 0, 0, 0, 0, ts, (128 - ts) / 4, 63 - (128 - ts) / 4});
  if (!mall && !stat && !fuse) for (int ts : {128, 124, 112}) cfgs.push_back({3, 3, 14, 7, 3, 0, 1, 0, 4, 0, 0, 0, 0, ts, (128 - ts) / 4, 63 - (128 - ts) / 4});
  for (const Cfg& c : cfgs) {
    const int nxr = c.nxo ? c.nxo : nx;
    a.nx = nxr; a.wpb = c.wpb; a.sync_rows = c.sync; a.work = c.work; a.stagger = c.stagger; a.tstride = c.tstride; a.vlo = c.vlo; a.vhi = c.vhi; a.ntiles = (ny + c.tstride - 1) / c.tstride; a.km = 0.999999; a.ka = 1e-9;
    a.R = c.R; a.L = c.L; a.nchunks = (nxr + c.R - 1) / c.R; a.persistent = c.persistent; a.nt_loads = c.nt;
    const long waves = (long)a.nchunks * a.ntiles;
    const size_t lds = c.wps >= 8 ? 0 : (size_t)(160 * 1024 / c.wps) - 1024;   // caps blocks per CU = waves per SIMD
    unsigned blocks = (unsigned)((waves + c.wpb - 1) / c.wpb);
    if (c.persistent) blocks = std::min<unsigned>(blocks, 256u * c.wps);
    auto launch = [&]() {
#define GO(NI, NO, DD) hipLaunchKernelGGL((k_stream<NI, NO, DD>), dim3(blocks), dim3(64 * c.wpb), lds * c.wpb / 4, 0, a)
      if (c.nin == 4 && c.nout == 4) GO(4, 4, 1); else if (c.nin == 4 && c.nout == 1) GO(4, 1, 1); else if (c.nin == 4 && c.D == 1) GO(4, 3, 1); else if (c.nin == 4) GO(4, 3, 2);
      else if (c.nin == 3 && c.D == 1) GO(3, 3, 1); else if (c.nin == 3) GO(3, 3, 2);
      else if (c.D == 1) GO(2, 1, 1); else GO(2, 1, 2);
    };
    if (lds > 64 * 1024) {
#define ATTR(NI, NO, DD) hipFuncSetAttribute(reinterpret_cast<const void*>(k_stream<NI, NO, DD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
      ATTR(4, 4, 1); ATTR(4, 1, 1); ATTR(4, 3, 1); ATTR(4, 3, 2); ATTR(3, 3, 1); ATTR(3, 3, 2); ATTR(2, 1, 1); ATTR(2, 1, 2);
    }
    for (int w = 0; w < 3; ++w) launch();
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = 1e3 * ms / reps;
    const double alg = (double)(c.nin + c.nout) * nxr * ny * 8.0;
    const double req = ((double)c.nin * (c.R + c.L) / c.R + c.nout) * nxr * ny * 8.0;
    printf("tile stride %3d lanes %d-%d | stagger %2d work %3d | nx %d wpb %2d sync %d | in %d out %d R %2d L %2d waves/SIMD %d %s D %d %s: %7.1f us  alg %.2f TB/s  requested %.2f TB/s  (%ld waves%s)\n", c.tstride, c.vlo, c.vhi, c.stagger, c.work, nxr, c.wpb, c.sync, c.nin, c.nout, c.R, c.L,
           c.wps, c.persistent == 2 ? "static" : c.persistent ? "queue" : "grid ", c.D, c.nt ? "nt-loads" : "        ", us, alg / us * 1e-6, req / us * 1e-6, waves,
           hipGetLastError() == hipSuccess ? "" : " ERROR");
    fflush(stdout);
  }
  return 0;
}
