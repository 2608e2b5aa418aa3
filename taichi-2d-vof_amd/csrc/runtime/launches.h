// runtime/launches.h -- chunk-length heuristics and L<T>: one launch wrapper per kernel (grid shape, chunk length, arguments), through the single launch() helper
//
// Part of the host-side runtime of libvof2d_hip.so; included (once, in this order) by vof2d_api.hip:
// context.h, launches.h, schedule.h, comm.h, selftest.h.  Everything here has internal linkage.
#pragma once
#include "context.h"

namespace {

// ------------------------------------------------------------------ chunk lengths (launch geometry)
// Rows per wave chunk.  Every marching kernel trades lead-in / halo rows per chunk (re-read from
// HBM by the vertical neighbour) against the number of waves.  Two effects decide:
//  * residency rounds: a launch whose waves exceed what the chip holds at once (occupancy x 1024
//    SIMDs) by a little runs a nearly empty extra round (measured on k_jacobi_tb at 4096^2: 3010
//    waves 116 us, 3080 waves 158 us), so the chunk length is chosen to make the launch k full
//    rounds, k as small as the maximum chunk length allows;
//  * with few cells the critical path of one wave dominates, so chunks never exceed what keeps
//    one round's worth of waves busy (short chunks on small grids).
// Occupancy comes from the runtime's query for the actual kernel (it depends on the compiled
// register count); a 5 % margin absorbs the over-reporting noted in MI355X_MICROARCH.md.
// Used for the two register-heavy, long-lived-wave kernels (k_jacobi_tb: -15 us per step at
// 4096^2, k_momentum: -3 us); the HBM-bound kernels with short-lived waves measured best with the
// plain cells-per-wave rule (chunk_rows) and keep it.
// (blocks of `threads` threads that fit the chip at once)
template <typename K>
long resident_blocks(vof2d_ctx* h, K kernel, int threads) {
  std::map<const void*, long>& cache = h->occ_cache;
  const void* key = reinterpret_cast<const void*>(kernel);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int blocks_per_cu = 0;
  long cap = 6L * 256;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, kernel, threads, 0) == hipSuccess && blocks_per_cu > 0) {
    int cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    cap = (long)blocks_per_cu * cus;
  } else {
    (void)hipGetLastError();
  }
  cache[key] = cap;
  return cap;
}
template <typename K>
long resident_waves(vof2d_ctx* h, K kernel) {
  std::map<const void*, long>& cache = h->occ_cache;  // per handle (one host thread per handle)
  const void* key = reinterpret_cast<const void*>(kernel);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int blocks_per_cu = 0;
  long cap = 3L * 256 * 4;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, kernel, 256, 0) == hipSuccess && blocks_per_cu > 0) {
    int cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    if (blocks_per_cu > 8) blocks_per_cu = 8;  // 32 waves per CU
    cap = (long)blocks_per_cu * cus * 4;
  } else {
    (void)hipGetLastError();
  }
  cache[key] = cap;
  return cap;
}
int chunk_rows_fit(const vof2d_ctx* h, int ntiles, long capacity, int rmin, int rmax) {
  const long rows = h->g.ihi - h->g.ilo + 1;
  const long cap = capacity * 95 / 100;
  int R_out = rmin;
  for (int k = 1; k <= 64; ++k) {
    long chunks_max = k * cap / ntiles;
    if (chunks_max < 1) continue;
    long R = (rows + chunks_max - 1) / chunks_max;
    if (R <= rmax) { R_out = (int)(R < rmin ? rmin : R); break; }
  }
  if (getenv("VOF2D_DEBUG"))
    fprintf(stderr, "[vof2d] chunk_rows_fit: rows=%ld tiles=%d capacity=%ld -> R=%d (%ld waves)\n", rows, ntiles,
            capacity, R_out, ((rows + R_out - 1) / R_out) * ntiles);
  return R_out;
}
// cells-per-wave rule (~4096 waves, chunk length a power of two), used by the x sweep, whose 6
// lead-in rows per chunk want long chunks (16 rows at 4096^2: 143 us; 8 rows 157 us, 4 rows 200 us)
int chunk_rows(const vof2d_ctx* h, int ntiles, int rmin, int rmax) {
  const long rows = h->g.ihi - h->g.ilo + 1;
  long R = rows * ntiles / 4096;
  if (R < rmin) R = rmin;
  if (R > rmax) R = rmax;
  long P = 1;
  while (P * 2 <= R) P *= 2;
  return (int)(P < rmin ? rmin : P);
}
// The streaming kernels with at most one halo row per side (single-sweep Jacobi, y sweep, the
// per-verb kernels): very short chunks.  With the nontemporal hints on their single-use streams the
// halo rows of vertically adjacent chunks -- consecutive blocks, resident at the same time -- are
// L2 hits, and many short-lived waves balance better than few long ones: k_jacobi at 4096^2 fp64
// 64 us with 2-row chunks (1 row 72 us, 4 rows 65 us, 8 rows 69 us, 32 rows 73.5 us); y sweep 112 us
// with 1 row, 116 us with 2, 136 us with 16.
int pick_rows(const vof2d_ctx* h, int ntiles) {
  (void)ntiles;
  if (h->rows_override > 0) return h->rows_override;
  return 2;
}
inline unsigned blocks_rows(int rows, int ntiles, int R) {
  const long waves = (long)((rows + R - 1) / R) * ntiles;
  return (unsigned)((waves + 3) / 4);
}
inline unsigned blocks_for(const vof2d_ctx* h, int ntiles, int R) {
  const int rows = h->g.ihi - h->g.ilo + 1;
  const long chunks = (rows + R - 1) / R;
  const long waves = chunks * ntiles;
  return (unsigned)((waves + 3) / 4);
}

// The BS form of k_momentum (range-checked buffer stores, see store_buf_nt): a lane's V columns must be stored or
// skipped together (tiles start on odd columns, so ny must be even) and a field must fit the 32-bit byte offsets of
// a buffer instruction.  Everything else takes the form with exec-masked global stores.  (k_transport -- which moves
// its L2-miss traffic at 5.6 TB/s either way -- gained nothing from the same change and keeps its global stores.)
inline bool buffer_stores_ok(const vof2d_ctx* h) {
  return h->buf_stores && (h->g.ny % 2 == 0) && (size_t)h->field_elems * h->esz < ((size_t)1 << 31) - (1u << 20);
}

// ------------------------------------------------------------------ launches
// From this many cells on a full domain runs the pair kernels whatever it holds (the rule of vof_step, runtime/schedule.h).  Re-measured
// with the kernels of round 6 (profiles/r06_forms_sweep.txt, ms/step pairs / chains on the rising bubble, 2 % gas): fp64 2048^2 0.253 / 0.227,
// 2560^2 0.325 / 0.312, 3072^2 0.405 / 0.411, 4096^2 0.592 / 0.614; fp32 2560^2 0.199 / 0.190, 3072^2 0.246 / 0.243, 4096^2 0.352 / 0.365
// (round 5, before the branch-free division tier: ties at 4096^2, hence 20 M then); 5120^2 0.78-0.83 / 0.89-0.92, 8192^2 1.67-1.71 / 2.17-2.20
constexpr long kTmAlwaysCells = 16000000L;
constexpr long kTbPlanWaves = 16384;   // waves of a k_jacobi_tb launch the work plan can describe
enum KernelId { kMomentum = 0, kSetBC, kJacobi, kJacobiTB, kCorrect, kFctX, kFctY, kNormals, kKappa, kPredictor,
                kRhs, kOther, kTransport, kJacobiPair, kTM, kTMUV, NKERNELS };
const char* const kKernelNames[NKERNELS] = {"k_momentum", "k_set_bc", "k_jacobi", "k_jacobi_tb", "k_correct",
                                            "k_fct_x", "k_fct_y", "k_normals", "k_kappa", "k_predictor", "k_rhs",
                                            "other", "k_transport", "k_jacobi_pair", "k_tm", "k_tm_uv"};   // (k_tm_uv: the k_tm launch that also stores u, v -- the last of a batch)

// One place through which every kernel is launched.  In profiling mode the dispatch carries its
// own start/stop events (hipExtLaunchKernelGGL: the begin/end timestamps of the dispatch itself,
// no extra barrier packets), otherwise it is a plain launch.
template <typename... KArgs, typename... Args>
void launch_block(vof2d_ctx* h, int kid, void (*kernel)(KArgs...), dim3 grid, unsigned threads, size_t lds, Args... args) {
  if (h->timed >= 0 && h->timed < vof2d_ctx::kMaxTimed) {
    const int k = h->timed++;
    h->tkid[k] = kid;
    hipExtLaunchKernelGGL(kernel, grid, dim3(threads), lds, h->stream, h->tev[2 * k], h->tev[2 * k + 1], 0, args...);
  } else {
    hipLaunchKernelGGL(kernel, grid, dim3(threads), lds, h->stream, args...);
  }
}
// (blocks of four waves -- four adjacent tiles -- for every kernel but k_tm, whose block is a pair of waves)
template <typename... KArgs, typename... Args>
void launch(vof2d_ctx* h, int kid, void (*kernel)(KArgs...), dim3 grid, size_t lds, Args... args) {
  launch_block(h, kid, kernel, grid, 256u, lds, args...);
}

template <typename T>
struct L {
  static constexpr int V = VecWidth<T>::V;
  static Consts<T> C(vof2d_ctx* h) { return round_consts<T>(h->cd); }

  static void init_F(vof2d_ctx* h, int ic) {
    dim3 grid((h->g.ny + 2 + 255) / 256, h->g.row_hi - h->g.row_lo + 1);
    launch(h, kOther, k_init_F<T>, grid, 0, h->g, C(h), F_<T>(h, fF), F_<T>(h, fF2), ic, h->d.Lx, h->d.Ly,
           (int)(h->d.coord_cast_f32 || h->d.dtype == VOF_F32));
  }
  // own_rows_only: the row loop skips the halo rows of a strip (wall ghost rows are never halo)
  template <int MASK>
  static void set_bc(vof2d_ctx* h, bool own_rows_only = false) {
    const int nr = h->g.row_hi - h->g.row_lo + 1;
    const int n = nr > h->g.ny + 2 ? nr : h->g.ny + 2;
    const int r0 = (own_rows_only && !h->g.wall_lo) ? h->d.own_lo : h->d.row_lo;
    const int r1 = (own_rows_only && !h->g.wall_hi) ? h->d.own_hi : h->d.row_hi;
    launch(h, kSetBC, k_set_bc<T, MASK>, dim3((n + 255) / 256), 0, h->g, F_<T>(h, fU), F_<T>(h, fV), F_<T>(h, fF),
           F_<T>(h, fF2), F_<T>(h, fP), F_<T>(h, fRHO), r0, r1);
  }
  static void bc_F_cols(vof2d_ctx* h, T* F, int r0, int r1) {
    if (r1 < r0) return;
    launch(h, kSetBC, k_bc_F_cols<T>, dim3((r1 - r0 + 256) / 256), 0, h->g, F, r0, r1);
  }
  static void nu_rho(vof2d_ctx* h) {
    dim3 grid((h->g.ny + 2 + 255) / 256, h->g.row_hi - h->g.row_lo + 1);
    launch(h, kOther, k_nu_rho<T>, grid, 0, h->g, C(h), (const T*)F_<T>(h, fF), F_<T>(h, fRHO), F_<T>(h, fNU));
  }
  static void post(vof2d_ctx* h) {
    dim3 grid((h->g.ny + 2 + 255) / 256, h->g.row_hi - h->g.row_lo + 1);
    launch(h, kOther, k_post<T>, grid, 0, h->g, F_<T>(h, fF), F_<T>(h, fF2));
  }
  static void normals(vof2d_ctx* h) {
    const int R = pick_rows(h, h->g.ntj);
    launch(h, kNormals, k_normals<T, V>, dim3(blocks_for(h, h->g.ntj, R)), 0, h->g, C(h), (const T*)F_<T>(h, fF),
           F_<T>(h, fMX), F_<T>(h, fMY), R);
  }
  static void kappa(vof2d_ctx* h) {
    const int R = pick_rows(h, h->g.ntj);
    launch(h, kKappa, k_kappa<T, V>, dim3(blocks_for(h, h->g.ntj, R)), 0, h->g, C(h), (const T*)F_<T>(h, fMX),
           (const T*)F_<T>(h, fMY), F_<T>(h, fKAPPA), R);
  }
  template <bool STORED>
  static void predictor(vof2d_ctx* h) {
    const int R = pick_rows(h, h->g.ntj);
    launch(h, kPredictor, k_predictor<T, V, STORED>, dim3(blocks_for(h, h->g.ntj, R)), 0, h->g, C(h),
           (const T*)F_<T>(h, fU), (const T*)F_<T>(h, fV), (const T*)F_<T>(h, fKAPPA), (const T*)F_<T>(h, fF),
           (const T*)F_<T>(h, fRHO), (const T*)F_<T>(h, fNU), F_<T>(h, fUS), F_<T>(h, fVS), R);
  }
  // fused normals + kappa + predictor + rhs (vof_step only)
  // rows [first, last] of the predictor and the rhs (last < first: all computable rows)
  static void momentum(vof2d_ctx* h, bool virt = false, int adapt_par = -1, int first = 1, int last = 0) {
    constexpr int Wt = 64 * V, Ht = TileHalo::momentum, ST = Wt - 2 * Ht;   // must match the kernel
    const int ntt = (h->g.ny + ST - 1) / ST;
    if (last < first) { first = h->g.ilo; last = h->g.ihi; }
    // one residency round while that keeps the chunks short (strips, small grids); on large grids
    // several rounds of 14-row chunks beat one round of long ones (4096^2: 184 vs 195 us, 8192^2:
    // 665 vs 758 us) -- the halo rows of adjacent, simultaneously resident chunks are L2 hits
    const bool bs = buffer_stores_ok(h) && (h->buf_stores & 1);
    int R = h->mom_rows > 0 ? h->mom_rows : chunk_rows_fit(h, ntt, bs ? resident_waves(h, k_momentum<T, V, true>) : resident_waves(h, k_momentum<T, V, false>), 4, 64);
    if (h->mom_rows <= 0 && R > 32) R = 14;
    const TbPlan tp = tb_plan(h, adapt_par);   // (one extra block: the planner wave)
    const unsigned mom_blocks = blocks_rows(last - first + 1, ntt, R);
    if (bs)
      launch(h, kMomentum, k_momentum<T, V, true>, dim3(mom_blocks + (tp.masks ? 1u : 0u)), 0, h->g, C(h), (const T*)F_<T>(h, fF),
             (const T*)F_<T>(h, fU), (const T*)F_<T>(h, fV), F_<T>(h, fUS), F_<T>(h, fVS), F_<T>(h, fRHS), R, ntt,
             virt ? 1 : 0, tp, first, last);
    else
      launch(h, kMomentum, k_momentum<T, V, false>, dim3(mom_blocks + (tp.masks ? 1u : 0u)), 0, h->g, C(h), (const T*)F_<T>(h, fF),
             (const T*)F_<T>(h, fU), (const T*)F_<T>(h, fV), F_<T>(h, fUS), F_<T>(h, fVS), F_<T>(h, fRHS), R, ntt,
             virt ? 1 : 0, tp, first, last);
  }
  template <bool STORED>
  static void rhs(vof2d_ctx* h) {
    const int R = pick_rows(h, h->g.ntj);
    launch(h, kRhs, k_rhs<T, V, STORED>, dim3(blocks_for(h, h->g.ntj, R)), 0, h->g, C(h), (const T*)F_<T>(h, fUS),
           (const T*)F_<T>(h, fVS), (const T*)F_<T>(h, fF), (const T*)F_<T>(h, fRHO), F_<T>(h, fRHS), R);
  }
  // one sweep src -> dst
  template <bool RESID>
  static void jacobi(vof2d_ctx* h, int src, int dst) {
    const int R = pick_rows(h, h->g.ntj);
    launch(h, kJacobi, k_jacobi<T, V, 2, RESID>, dim3(blocks_for(h, h->g.ntj, R)), 0, h->g, C(h),
           (const T*)F_<T>(h, src), (const T*)F_<T>(h, fRHS), F_<T>(h, dst), R, h->d_courant + 1);
  }
  // TS sweeps src -> dst in one launch, with VV columns per lane
  template <int TS, int VV>
  static int jacobi_tb_plan(vof2d_ctx* h, bool sq, int& ntt) {
    constexpr int Wt = 64 * VV;
    const int Ht = ((TS - 1 + (sq ? 1 : 0) + VV - 1) / VV) * VV, ST = Wt - 2 * Ht;  // must match the kernel
    ntt = (h->g.ny + ST - 1) / ST;
    const long cap = sq ? resident_waves(h, k_jacobi_tb<T, VV, TS, true, false>) : resident_waves(h, k_jacobi_tb<T, VV, TS, false, false>);
    return h->tb_rows > 0 ? h->tb_rows : chunk_rows_fit(h, ntt, cap, 4, 96);
  }
  // k_jacobi_pair (two five-sweep launches as one, kernels/jacobi_pair.h): square cells, ten sweeps per step at least
  static bool jacobi_pair_ok(vof2d_ctx* h) {
    const Consts<T> cc = C(h);
    // (both precisions since the chained batches: 4096^2 fp32 dam-break 0.268 ms/step in the k_tm form with the pairs, 0.343 in
    //  chains; knob values 1 and 2 are the same now)
    return h->jpair >= 1 && cc.dxi2 == cc.dyi2 && !h->tb_general && h->tb >= 5 &&
           h->d.jacobi_iters % 10 == 0;
  }
  // Columns per lane of the pair kernels: 2, or -- fp32, knob "pair_vec4" -- 4: a lane then moves the 16 bytes per row the fp64
  // kernels move, a tile is 256 columns (232 / 240 of them stored instead of 108 / 112 of 128), and the cross-lane moves and the
  // scalar bookkeeping of a row serve twice the cells.
  // EXPERIMENT (make variant NAME=vec4 EXTRA=-DVOF_PAIR_VEC4, then knob pair_vec4 = 1; tools/probes/forms_ab.py): same values,
  // 128 / 135-149 VGPRs (four / three waves per SIMD) -- and 4096^2 fp32 0.62 ms/step against 0.35 with two columns per lane
  // and 0.33 for the chains: a pair's step lasts as long as its longer wave's instructions, and a wave now carries twice as
  // many.  The product does not instantiate it.
  static int pair_vec(const vof2d_ctx* h) {
#ifdef VOF_PAIR_VEC4
    return (sizeof(T) == 4 && h->pair_vec4 && h->g.ny % 4 == 0 && buffer_stores_ok(h)) ? 4 : V;
#else
    (void)h;
    return V;
#endif
  }
  template <int VV>
  static int jacobi_pair_geom_v(vof2d_ctx* h, int& ntt) {
    constexpr int ST = 64 * VV - 2 * (((2 * 5 + VV - 1) / VV) * VV);   // must match the kernel: 108 columns (232 with four per lane)
    ntt = (h->g.ny + ST - 1) / ST;
    const bool bs = buffer_stores_ok(h) && (h->buf_stores & 2);
    const long cap = bs ? resident_blocks(h, k_jacobi_pair<T, VV, 5, true>, 128) : resident_blocks(h, k_jacobi_pair<T, VV, 5, false>, 128);
    return h->jpair_rows > 0 ? h->jpair_rows : chunk_rows_fit(h, ntt, cap, 8, 160);
  }
  static int jacobi_pair_geom(vof2d_ctx* h, int& ntt) {
#ifdef VOF_PAIR_VEC4
    if constexpr (sizeof(T) == 4) { if (pair_vec(h) == 4) return jacobi_pair_geom_v<4>(h, ntt); }
#endif
    return jacobi_pair_geom_v<V>(h, ntt);
  }
  // ten sweeps src -> dst
  template <int VV>
  static void jacobi_pair_v(vof2d_ctx* h, int src, int dst, int adapt_par, int first, int last) {
    int ntt = 0;
    const int R = jacobi_pair_geom_v<VV>(h, ntt);
    const TbPlan tp = tb_plan(h, adapt_par);
    const unsigned pairs = tp.masks ? (unsigned)tp.waves : (unsigned)(((last - first + R) / R) * ntt);
    const bool bs = buffer_stores_ok(h) && (h->buf_stores & 2);
    if (bs)
      launch_block(h, kJacobiPair, k_jacobi_pair<T, VV, 5, true>, dim3(pairs), 128u, 0, h->g, C(h), (const T*)F_<T>(h, src),
                   (const T*)F_<T>(h, fRHS), F_<T>(h, dst), R, ntt, tp, first, last);
    else
      launch_block(h, kJacobiPair, k_jacobi_pair<T, VV, 5, false>, dim3(pairs), 128u, 0, h->g, C(h), (const T*)F_<T>(h, src),
                   (const T*)F_<T>(h, fRHS), F_<T>(h, dst), R, ntt, tp, first, last);
  }
  static void jacobi_pair(vof2d_ctx* h, int src, int dst, int adapt_par = -1, int first = 1, int last = 0) {
    if (last < first) { first = h->g.ilo; last = h->g.ihi; }
#ifdef VOF_PAIR_VEC4
    if constexpr (sizeof(T) == 4) { if (pair_vec(h) == 4) return jacobi_pair_v<4>(h, src, dst, adapt_par, first, last); }
#endif
    jacobi_pair_v<V>(h, src, dst, adapt_par, first, last);
  }
  // the work plan of the step's five-sweep launches (see tb_make_plan): active on parity-keyed step
  // sequences (adapt_par = istep & 1), square or not, two columns per lane, up to TB_COLS tile columns
  static TbPlan tb_plan(vof2d_ctx* h, int adapt_par) {
    TbPlan tp{nullptr, nullptr, 0, 0, 0, 0, 0};
    if (adapt_par < 0 || !h->tb_adapt || h->tb < 5 || h->tb_rows > 0) return tp;
    const Consts<T> cc = C(h);
    const bool sq = cc.dxi2 == cc.dyi2 && !h->tb_general;
    int ntt = 0;
    int R;
    long waves;
    if (h->jpair_active) {   // the step's Jacobi launches are k_jacobi_pair's: the plan's "waves" are pairs on 108-column tiles (jacobi_pair_geom); a plan is only ever read by the kernel it was planned for -- enqueue_tm_head, tm5_head
      R = jacobi_pair_geom(h, ntt);
      waves = (long)((h->g.ihi - h->g.ilo + R) / R) * ntt;
    } else {
      R = jacobi_tb_plan<5, V>(h, sq, ntt);
      waves = (long)blocks_for(h, ntt, R) * 4;
    }
    if (ntt > TB_COLS || waves > kTbPlanWaves) return tp;
    tp.masks = h->d_tbmask;
    tp.plan = h->d_tbmask + 2 * TB_BANDS * (TB_COLS / 64);
    tp.ntt = ntt; tp.R = R; tp.waves = (int)waves; tp.par = adapt_par;
    tp.slow10 = h->jpair_active ? h->pair_slow10 : h->tb_slow10;   // (what a row of a reported band costs: per kernel)
    return tp;
  }
  template <int TS, int VV>
  static void jacobi_tb_launch(vof2d_ctx* h, const Consts<T>& cc, bool sq, int src, int dst, int R, int ntt, int adapt_par = -1,
                               int first = 1, int last = 0) {
    if (last < first) { first = h->g.ilo; last = h->g.ihi; }

    unsigned long long* none = nullptr;
    TbPlan tp{nullptr, nullptr, 0, 0, 0, 0, 0};
    if (TS == 5 && VV == V) tp = tb_plan(h, adapt_par);
    // (with a plan the launch holds the waves of the whole grid's plan, whatever part of the rows it is for: every
    // wave takes the part of its planned chunk inside [first, last], or nothing)
    const unsigned nblk = tp.masks ? blocks_for(h, ntt, R) : blocks_rows(last - first + 1, ntt, R);
    // the buffer-store form where the launch is ONE residency round of the chunk plan (4096^2, the strips of a multi-GPU
    // run: 105 -> 103 us, 53.7 -> 49.0 us): it needs 126 VGPRs instead of 129, i.e. four waves per SIMD are resident
    // where the plan counted on three, which breaks the round structure of a multi-round launch (8192^2 on one GPU:
    // 397 -> 435 us)
    const bool one_round = (long)blocks_for(h, ntt, R) * 4 <= resident_waves(h, k_jacobi_tb<T, VV, TS, true, false>);
    constexpr bool kBufForm = sizeof(T) * VV == 16 || sizeof(T) * VV == 8;   // (store_buf_nt: one b128 / b64 store per lane)
    if (kBufForm && sq && one_round && buffer_stores_ok(h) && (h->buf_stores & 2)) {
      if constexpr (kBufForm)
        launch(h, kJacobiTB, k_jacobi_tb<T, VV, TS, true, false, true>, dim3(nblk), 0, h->g, cc,
               (const T*)F_<T>(h, src), (const T*)F_<T>(h, fRHS), F_<T>(h, dst), R, ntt, none, tp, first, last);
    } else if (sq)
      launch(h, kJacobiTB, k_jacobi_tb<T, VV, TS, true, false>, dim3(nblk), 0, h->g, cc,
             (const T*)F_<T>(h, src), (const T*)F_<T>(h, fRHS), F_<T>(h, dst), R, ntt, none, tp, first, last);
    else
      launch(h, kJacobiTB, k_jacobi_tb<T, VV, TS, false, false>, dim3(nblk), 0, h->g, cc,
             (const T*)F_<T>(h, src), (const T*)F_<T>(h, fRHS), F_<T>(h, dst), R, ntt, none, tp, first, last);
  }
  // TS sweeps src -> dst, the last of which also reduces max|p_new - p| and max|p_new| over the owned
  // rows into d_courant[1..2] (the residual-terminated solve): same values as
  // jacobi_tb<TS>, same launch plan (the RESID instantiation needs a few registers more, so its
  // own occupancy decides the chunk length)
  template <int TS>
  static void jacobi_tb_resid(vof2d_ctx* h, int src, int dst) {
    const Consts<T> cc = C(h);
    const bool sq = cc.dxi2 == cc.dyi2 && !h->tb_general;
    constexpr int Wt = 64 * V;
    const int Ht = ((TS - 1 + (sq ? 1 : 0) + V - 1) / V) * V, ST = Wt - 2 * Ht;
    const int ntt = (h->g.ny + ST - 1) / ST;
    const long cap = sq ? resident_waves(h, k_jacobi_tb<T, V, TS, true, true>) : resident_waves(h, k_jacobi_tb<T, V, TS, false, true>);
    const int R = h->tb_rows > 0 ? h->tb_rows : chunk_rows_fit(h, ntt, cap, 4, 96);
    const TbPlan notp{nullptr, nullptr, 0, 0, 0, 0};   // uniform layout
    if (sq)
      launch(h, kJacobiTB, k_jacobi_tb<T, V, TS, true, true>, dim3(blocks_for(h, ntt, R)), 0, h->g, cc,
             (const T*)F_<T>(h, src), (const T*)F_<T>(h, fRHS), F_<T>(h, dst), R, ntt, h->d_courant + 1, notp, h->g.ilo, h->g.ihi);
    else
      launch(h, kJacobiTB, k_jacobi_tb<T, V, TS, false, true>, dim3(blocks_for(h, ntt, R)), 0, h->g, cc,
             (const T*)F_<T>(h, src), (const T*)F_<T>(h, fRHS), F_<T>(h, dst), R, ntt, h->d_courant + 1, notp, h->g.ilo, h->g.ihi);
  }
  template <int TS>
  static void jacobi_tb(vof2d_ctx* h, int src, int dst, int adapt_par = -1, int first = 1, int last = 0) {
    const Consts<T> cc = C(h);
    const bool sq = cc.dxi2 == cc.dyi2 && !h->tb_general;  // square cells: the product-carrying pipeline
    int ntt = 0;
#ifdef VOF_TB_VV      // EXPERIMENT: wider tiles for the fused Jacobi kernel
    if (TS == 5 && sizeof(T) == 8) {
      const int R = jacobi_tb_plan<TS, VOF_TB_VV>(h, sq, ntt);
      jacobi_tb_launch<TS, VOF_TB_VV>(h, cc, sq, src, dst, R, ntt, -1, first, last);
      return;
    }
#endif
    const int R = jacobi_tb_plan<TS, V>(h, sq, ntt);
    jacobi_tb_launch<TS, V>(h, cc, sq, src, dst, R, ntt, adapt_par, first, last);
  }
  template <bool STORED>
  static void correct(vof2d_ctx* h) {
    const int R = pick_rows(h, h->g.ntj);
    launch(h, kCorrect, k_correct<T, V, STORED>, dim3(blocks_for(h, h->g.ntj, R)), 0, h->g, C(h),
           (const T*)F_<T>(h, fP), (const T*)F_<T>(h, fF), (const T*)F_<T>(h, fRHO), (const T*)F_<T>(h, fUS),
           (const T*)F_<T>(h, fVS), F_<T>(h, fU), F_<T>(h, fV), R, h->d_courant);
  }
  // sweeps read fld[fF], write fld[fF2]; the caller swaps the two afterwards.
  // CORR: the sweep also performs update_uv (reads u*, v*, p; writes u, v) -- see k_fct_x.
  // rows [first, last] of the sweep's output (0, 0: all computable rows)
  template <bool POST, bool CORR>
  static void fct_x(vof2d_ctx* h, int first = 0, int last = 0) {
    if (first == 0 && last == 0) { first = h->g.ilo; last = h->g.ihi; }
    const int forced = CORR && h->fctx_corr_rows > 0 ? h->fctx_corr_rows : h->fctx_rows;
    const int R = forced > 0 ? forced : chunk_rows(h, h->g.ntj, 4, 16);
    launch(h, kFctX, k_fct_x<T, V, POST, CORR>, dim3(blocks_rows(last - first + 1, h->g.ntj, R)), 0, h->g, C(h),
           (const T*)F_<T>(h, fF), (const T*)F_<T>(h, fU), F_<T>(h, fF2), R, (const T*)F_<T>(h, fUS),
           (const T*)F_<T>(h, fVS), (const T*)F_<T>(h, fP), F_<T>(h, fU), F_<T>(h, fV), h->d_courant, first, last);
  }
  template <bool POST, bool CORR>
  static void fct_y(vof2d_ctx* h, int first = 0, int last = 0) {
    if (first == 0 && last == 0) { first = h->g.ilo; last = h->g.ihi; }
    const int R = h->rows_override > 0 ? h->rows_override : 1;   // rows are independent in this sweep
    launch(h, kFctY, k_fct_y<T, V, POST, CORR>, dim3(blocks_rows(last - first + 1, h->nty, R)), 0, h->g, C(h),
           (const T*)F_<T>(h, fF), (const T*)F_<T>(h, fV), F_<T>(h, fF2), R, h->nty, (const T*)F_<T>(h, fUS),
           (const T*)F_<T>(h, fVS), (const T*)F_<T>(h, fP), F_<T>(h, fU), F_<T>(h, fV), h->d_courant, first, last);
  }
  // k_transport of this step + k_momentum of the next in one launch (k_tm): reads fld[fF], fld[fUS], fld[fVS], fld[fP];
  // writes fld[fF2], rhs, u*' / v*' into fld[fMX] / fld[fMY] (the caller alternates the pairs and swaps F), u and v
  // only with STORE_UV; adapt_par: parity of the NEXT step (its planner block rides here as it does in k_momentum)
  static int tm_chunk_rows(const vof2d_ctx* h, long rows, int ntf, long cap) {
    if (h->tm_rows > 0) return h->tm_rows;
    long k = (rows * ntf + cap * 25) / (cap * 50);
    if (k < 2) k = 2;
    long chunks = k * cap * 97 / 100 / ntf;
    if (chunks < 1) chunks = 1;
    const long R = (rows + chunks - 1) / chunks;
    return (int)(R < 16 ? 16 : (R > 96 ? 96 : R));
  }
  template <bool YFIRST, bool STORE_UV>
  static void tm(vof2d_ctx* h, int adapt_par, int first = 1, int last = 0, int rows_forced = 0, int first2 = 1, int last2 = 0) {
    if (last < first) { first = h->g.ilo; last = h->g.ihi; }
#ifdef VOF_PAIR_VEC4
    if constexpr (sizeof(T) == 4) { if (pair_vec(h) == 4) return tm_v<4, YFIRST, STORE_UV>(h, adapt_par, first, last, rows_forced, first2, last2); }
#endif
    tm_v<V, YFIRST, STORE_UV>(h, adapt_par, first, last, rows_forced, first2, last2);
  }
  template <int VV, bool YFIRST, bool STORE_UV>
  static void tm_v(vof2d_ctx* h, int adapt_par, int first, int last, int rows_forced, int first2, int last2) {
    constexpr int ST = 64 * VV - 2 * TmGeom::HF;
    const int ntf = (h->g.ny + ST - 1) / ST;
    // pair chunks: a whole number of residency rounds, just filled (6 pairs per CU: 24 KB of LDS each) -- a launch that needs a
    // little more than k rounds pays for k + 1 --, as many rounds as keep the chunks near 50 rows (one round of 100-row
    // chunks: every step of every pair takes 3.3 us instead of 1.9).  4096^2, 112-column tiles, us per launch: 40 rows
    // (2.5 rounds) 261 / 281 (inside / behind the front), 48 252 / 282, 52 253 / 280, 54 256 / 277, 56 257 / 284,
    // 100 376 / 381 (tools/probes/pair_bound.py --rows)
    const int R = rows_forced > 0 ? rows_forced : tm_chunk_rows(h, last - first + 1, ntf, resident_blocks(h, k_tm<T, VV, YFIRST, STORE_UV, true>, 128));
    const TbPlan tp = tb_plan(h, adapt_par);
    const unsigned pairs = (unsigned)((((last - first + R) / R) + (last2 >= first2 ? (last2 - first2 + R) / R : 0)) * ntf) + (tp.masks ? 1u : 0u);
    const bool bs = buffer_stores_ok(h) && (h->buf_stores & 4);
    if (bs)
      launch_block(h, STORE_UV ? kTMUV : kTM, k_tm<T, VV, YFIRST, STORE_UV, true>, dim3(pairs), 128u, 0, h->g, C(h), (const T*)F_<T>(h, fF), F_<T>(h, fF2), ntf,
             (const T*)F_<T>(h, fUS), (const T*)F_<T>(h, fVS), (const T*)F_<T>(h, fP), F_<T>(h, fU), F_<T>(h, fV),
             F_<T>(h, fMX), F_<T>(h, fMY), F_<T>(h, h->tm_rhs_alt ? fKAPPA : fRHS), h->d_courant, R, tp, first, last, first2, last2);
    else
      launch_block(h, STORE_UV ? kTMUV : kTM, k_tm<T, VV, YFIRST, STORE_UV, false>, dim3(pairs), 128u, 0, h->g, C(h), (const T*)F_<T>(h, fF), F_<T>(h, fF2), ntf,
             (const T*)F_<T>(h, fUS), (const T*)F_<T>(h, fVS), (const T*)F_<T>(h, fP), F_<T>(h, fU), F_<T>(h, fV),
             F_<T>(h, fMX), F_<T>(h, fMY), F_<T>(h, h->tm_rhs_alt ? fKAPPA : fRHS), h->d_courant, R, tp, first, last, first2, last2);
  }
  // update_uv + both sweeps + post_process_f in one pass (k_transport); reads fld[fF], writes fld[fF2]
  static int transport_rows(const vof2d_ctx* h) {
    return h->fctx_corr_rows > 0 ? h->fctx_corr_rows : chunk_rows(h, h->nty, 4, 16);
  }
  static long range_chunks(const RowRanges& rr) {
    long n = 0;
    for (int k = 0; k < 3; ++k)
      if (rr.last[k] >= rr.first[k]) n += (rr.last[k] - rr.first[k] + rr.R[k]) / rr.R[k];
    return n;
  }
  // the rows of rr (all computable rows by default)
  template <bool YFIRST>
  static void transport(vof2d_ctx* h, const RowRanges* ranges = nullptr) {
    RowRanges rr;
    if (ranges) rr = *ranges;
    else rr = RowRanges{{h->g.ilo, 1, 1}, {h->g.ihi, 0, 0}, {transport_rows(h), 1, 1}};
    const unsigned tr_blocks = (unsigned)((range_chunks(rr) * h->nty + 3) / 4);
    launch(h, kTransport, k_transport<T, V, YFIRST>, dim3(tr_blocks), 0, h->g, C(h),
           (const T*)F_<T>(h, fF), F_<T>(h, fF2), h->nty, (const T*)F_<T>(h, fUS), (const T*)F_<T>(h, fVS),
           (const T*)F_<T>(h, fP), F_<T>(h, fU), F_<T>(h, fV), h->d_courant, rr);
  }
};

}  // namespace
