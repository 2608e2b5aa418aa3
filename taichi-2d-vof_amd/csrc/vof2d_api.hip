// vof2d_api.hip -- C ABI (include/vof2d.h) over the gfx950 kernels.
//
// Host-side runtime of the drop-in: owns the device arena, the HIP stream, the
// per-step launch schedule (eager or hipGraph replay) and the pitched
// host<->device copies behind to_numpy()/from_numpy().  No CPU compute path
// exists here: every verb is a kernel launch.
#include <float.h>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>

#include "../../include/vof2d.h"
#include "vof2d_kernels.h"

using namespace vof;

struct RcclId { char internal[VOF_COMM_ID_BYTES]; };  // ncclUniqueId, passed by value to ncclCommInitRank

namespace {

enum FieldId { fF = 0, fF2, fU, fV, fP, fPT, fUS, fVS, fMX, fMY, fKAPPA, fRHO, fNU, fRHS, NFIELDS };
const char* const kFieldNames[NFIELDS] = {"F", "F2", "u", "v", "p", "pt", "u_star", "v_star",
                                          "mx", "my", "kappa", "rho", "nu", "rhs"};

struct ConstsD {  // Python-double values
  double dt, dx, dy, dxi, dyi, dxi2, dyi2, rho_l, rho_g, nu_l, nu_g, sigma, gx, gy;
  double nrm_x, nrm_y, kap_x, kap_y, dxdy, dtdy, dtdx, cfl_x, cfl_y, half_dx, half_dy, sqrt2dx, tiny;
  double ic1_x2, ic1_y2, ic_r, ic_cx, ic2_cy, ic3_cy, ic3_pool;
};

double host_node_coord(double L, int n, int k, int cast_f32) {
  // k-th entry of hstack((0, linspace(0, L, n+1), L)) [.astype(float32)], 2dvof.py:43-46
  double v = k == 0 ? 0.0 : (k >= n + 1 ? L : (double)(k - 1) * (L / (double)n));
  if (cast_f32) v = (double)(float)v;
  return v;
}

void compute_consts(const vof2d_desc& d, ConstsD& c) {
  const int cast = d.coord_cast_f32 || d.dtype == VOF_F32;  // an f32 field rounds the coordinates anyway
  // 2dvof.py:47-50: Python-scope reads of x[imin+2], x[imin+1] -> Python doubles
  const double dx = host_node_coord(d.Lx, d.nx, 3, cast) - host_node_coord(d.Lx, d.nx, 2, cast);
  const double dy = host_node_coord(d.Ly, d.ny, 3, cast) - host_node_coord(d.Ly, d.ny, 2, cast);
  const double dxi = 1 / dx, dyi = 1 / dy;
  c.dt = d.dt; c.dx = dx; c.dy = dy; c.dxi = dxi; c.dyi = dyi;
  c.dxi2 = std::pow(dxi, 2.0);  // dxi ** 2  (:216)
  c.dyi2 = std::pow(dyi, 2.0);
  c.rho_l = d.rho_l; c.rho_g = d.rho_g; c.nu_l = d.nu_l; c.nu_g = d.nu_g;
  c.sigma = d.sigma; c.gx = d.gx; c.gy = d.gy;
  c.nrm_x = -1 / (2 * dx);  // :287
  c.nrm_y = -1 / (2 * dy);
  c.kap_x = 1 / dx / 2;     // :308
  c.kap_y = 1 / dy / 2;
  c.dxdy = dx * dy;         // :324
  c.dtdy = d.dt * dy;       // :324
  c.dtdx = d.dt * dx;       // :388
  c.cfl_x = 0.25 * dx;      // :274
  c.cfl_y = 0.25 * dy;      // :279
  c.half_dx = dx / 2;       // :105
  c.half_dy = dy / 2;
  c.sqrt2dx = std::sqrt(2.0) * dx;  // :131
  c.tiny = 1e-10;           // :300
  c.ic1_x2 = d.Lx / 3;      // :141
  c.ic1_y2 = d.Ly / 2;      // :143
  c.ic_r = d.Lx / 12;       // :150
  c.ic_cx = d.Lx / 2;       // :151
  c.ic2_cy = 2 * (d.Lx / 12);         // :151
  c.ic3_cy = d.Ly - 3 * (d.Lx / 12);  // :155
  c.ic3_pool = d.Ly * 0.37;           // :157
}

template <typename T>
Consts<T> round_consts(const ConstsD& s) {
  Consts<T> c;
#define R1(n) c.n = (T)s.n
  R1(dt); R1(dx); R1(dy); R1(dxi); R1(dyi); R1(dxi2); R1(dyi2); R1(rho_l); R1(rho_g); R1(nu_l); R1(nu_g);
  R1(sigma); R1(gx); R1(gy); R1(nrm_x); R1(nrm_y); R1(kap_x); R1(kap_y); R1(dxdy); R1(dtdy); R1(dtdx);
  R1(cfl_x); R1(cfl_y); R1(half_dx); R1(half_dy); R1(sqrt2dx); R1(tiny);
  // RN(1/b) in T arithmetic for div_by_const
  c.inv_dx = (T)1 / c.dx; c.inv_dy = (T)1 / c.dy; c.inv_dt = (T)1 / c.dt; c.inv_dxdy = (T)1 / c.dxdy;
  R1(ic1_x2); R1(ic1_y2); R1(ic_r); R1(ic_cx); R1(ic2_cy); R1(ic3_cy); R1(ic3_pool);
#undef R1
  return c;
}

// div_by_const (vof2d_kernels.h) returns the correctly rounded a / b from y = RN(1/b) for every
// denominator except one whose significand is all ones (Markstein).  The denominators it is used
// with are a handful of constants; refuse the (practically impossible) bad ones at creation.
template <typename T>
bool all_ones_significand(T b) {
  if (sizeof(T) == 8) {
    uint64_t u;
    double d = (double)b;
    memcpy(&u, &d, 8);
    return (u & 0xFFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFFull;
  }
  uint32_t u;
  float f = (float)b;
  memcpy(&u, &f, 4);
  return (u & 0x7FFFFFu) == 0x7FFFFFu;
}
template <typename T>
bool divisors_ok(const ConstsD& s) {
  const Consts<T> c = round_consts<T>(s);
  if (all_ones_significand(c.dx) || all_ones_significand(c.dy) || all_ones_significand(c.dt) ||
      all_ones_significand(c.dxdy))
    return false;
  for (int e = 0; e <= 2; ++e)      // ap = -(ae + aw + an + as), each term present or 0 (2dvof.py:258-262)
    for (int n = 0; n <= 2; ++n) {
      if (e + n == 0) continue;
      T ap = (T)0;
      for (int k = 0; k < e; ++k) ap = ap + c.dxi2;
      for (int k = 0; k < n; ++k) ap = ap + c.dyi2;
      if (all_ones_significand(ap)) return false;
    }
  return true;
}

}  // namespace

struct vof2d_ctx {
  vof2d_desc d;
  ConstsD cd;
  Geom g;
  int V;          // elements per lane
  size_t esz;     // sizeof(T)
  int nty;        // y-sweep tiles
  size_t field_elems;
  char* arena = nullptr;
  void* fld[NFIELDS];
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int device = 0;
  unsigned long long* d_courant = nullptr;  // device counters: [0] courant, [1] max|p_new - p| bits, [2] max|p_new| bits (residual solve)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int64_t istep = 0;
  int rows_override = 0;
  int tb = 5;           // Jacobi sweeps fused per launch (1 = plain kernel)
  int tb_rows = 0;      // rows per wave chunk of the fused kernel (0 = heuristic)
  int mom_rows = 0;     // rows per wave chunk of k_momentum (0 = heuristic)
  int tb_general = 0;   // force the general (dx != dy) fused Jacobi kernel
  int tb_narrow = 0;    // 1: the one-column-per-lane fused Jacobi kernel on thin, wide strips (see jacobi_tb)
  int tb_wide = 0;      // fp32: four columns per lane (16-byte loads, 256-column tiles) in the fused Jacobi kernel
  int tb_adapt = 1;     // fused steps: shorter chunks on the tile columns the tiny-value front is crossing (k_jacobi_tb)
  unsigned long long* d_tbmask = nullptr;  // work plan of k_jacobi_tb (TbPlan): 2 x TB_BANDS mask words, then the plan (1 + waves entries)
  long tbplan_cap = 0;                     // waves the plan area holds
  int fctx_rows = 0;    // rows per wave chunk of k_fct_x (0 = heuristic, at most 16)
  int fctx_corr_rows = 0;  // ... of its update_uv-carrying form (0 = same rule)
  int fuse_momentum = 1;
  int fuse_correct = 1; // vof_step on a full domain: update_uv inside the first FCT sweep
  int fuse_transport = 1;  // ... and both FCT sweeps in one kernel (k_transport), full domains only
  int band_rows = 4;       // rows per wave chunk of the edge-band launch of the fused transport (strips)
  int virtual_ghosts = 1;  // ... without the step's set_BC launch (k_momentum forms the ghost cells it reads)
  void* f_home = nullptr;  // the buffer fld[fF] pointed to at creation (orientation of the F / twin pair)
  int phase_graph_ori = 0; // orientation the gphase / gxchg graphs were captured in
  hipGraphExec_t gexec[2][2] = {};  // whole step, [istep parity][F in its home buffer ? 0 : 1]
  hipGraphExec_t gphase[9] = {};  // phase 0, then phases 1..4 x istep parity (slot 2 * phase - 1 + parity)
  int next_phase = 0;
  bool f_ghosts_dirty = true;  // F's ghost cells may not satisfy set_BC (after set_init_F / from_numpy / a single verb)
  bool uv_ghosts_dirty = false; // u / v were written without a set_BC since (update_uv verb, from_numpy): their ghost cells are not mirror images
  bool ghosts_virtual = false; // the last fused step skipped its set_BC launch: the ghost cells in memory are stale
                               // (k_momentum forms the ones it reads; everything else goes through settle_ghosts)
  void* vis = nullptr;      // scratch for the display fields (vof_get_vis_field / vof_interp_velocity)
  size_t vis_bytes = 0;
  // built-in in-situ profiler (vof_profile_steps): every launch carries a start/stop event pair
  static constexpr int kMaxTimed = 96;
  hipEvent_t tev[2 * kMaxTimed] = {};
  int timed = -1;             // -1: off; otherwise launches recorded in the current batch
  int tkid[kMaxTimed];        // kernel id of each recorded launch
  double prof_sum_ms[16] = {};
  long prof_cnt[16] = {};
  std::map<const void*, long> occ_cache;  // resident waves per kernel function (resident_waves)
  // strip halo exchange over RCCL (vof_comm_init): own communicator, stream and events
  void* comm = nullptr;          // ncclComm_t
  hipStream_t cstream = nullptr; // RCCL's kernels run here, next to the compute stream
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  hipEvent_t ev_fork[3] = {nullptr, nullptr, nullptr};  // one per exchange of a step (graph capture forks)
  hipGraphExec_t gxchg[2][5][2] = {};   // whole step + exchanges, [istep parity][overlap mode][F / twin orientation]
  int xchg_graph = 1;                // 0 after a failed capture (or VOF2D_XCHG_GRAPH=0): eager launches
  int64_t xchg_steps = 0;            // steps run by vof_step_exchange (the first one is always eager)
  int64_t xchg_graph_steps = 0;      // ... of which replayed from a captured graph
  double* d_red = nullptr;           // device scalar of vof_comm_allreduce_max
  int comm_rank = 0, comm_world = 1;
  int peer_lo = -1, peer_hi = -1;  // ranks owning the rows below own_lo / above own_hi (-1: wall)
  char err[512];
};

namespace {

#define HIPCHK(h, call)                                                                          \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      snprintf((h)->err, sizeof((h)->err), "%s:%d %s -> %s", __FILE__, __LINE__, #call,          \
               hipGetErrorString(e_));                                                           \
      return VOF_EHIP;                                                                           \
    }                                                                                            \
  } while (0)

int fail(vof2d_ctx* h, int code, const char* msg) {
  if (h) snprintf(h->err, sizeof(h->err), "%s", msg);
  return code;
}

int field_id(const char* name) {
  if (!name) return -1;
  for (int k = 0; k < NFIELDS; ++k)
    if (!strcmp(name, kFieldNames[k])) return k;
  return -1;
}

template <typename T> T* F_(vof2d_ctx* h, int id) { return reinterpret_cast<T*>(h->fld[id]); }

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

// ------------------------------------------------------------------ launches
constexpr long kTbPlanWaves = 16384;   // waves of a k_jacobi_tb launch the work plan can describe
enum KernelId { kMomentum = 0, kSetBC, kJacobi, kJacobiTB, kCorrect, kFctX, kFctY, kNormals, kKappa, kPredictor,
                kRhs, kOther, kTransport, NKERNELS };
const char* const kKernelNames[NKERNELS] = {"k_momentum", "k_set_bc", "k_jacobi", "k_jacobi_tb", "k_correct",
                                            "k_fct_x", "k_fct_y", "k_normals", "k_kappa", "k_predictor", "k_rhs",
                                            "other", "k_transport"};

// One place through which every kernel is launched.  In profiling mode the dispatch carries its
// own start/stop events (hipExtLaunchKernelGGL: the begin/end timestamps of the dispatch itself,
// no extra barrier packets), otherwise it is a plain launch.
template <typename... KArgs, typename... Args>
void launch(vof2d_ctx* h, int kid, void (*kernel)(KArgs...), dim3 grid, size_t lds, Args... args) {
  if (h->timed >= 0 && h->timed < vof2d_ctx::kMaxTimed) {
    const int k = h->timed++;
    h->tkid[k] = kid;
    hipExtLaunchKernelGGL(kernel, grid, dim3(256), lds, h->stream, h->tev[2 * k], h->tev[2 * k + 1], 0, args...);
  } else {
    hipLaunchKernelGGL(kernel, grid, dim3(256), lds, h->stream, args...);
  }
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
  static void momentum(vof2d_ctx* h, bool virt = false, int adapt_par = -1) {
    constexpr int Wt = 64 * V, Ht = ((2 + V - 1) / V) * V, ST = Wt - 2 * Ht;
    const int ntt = (h->g.ny + ST - 1) / ST;
    // one residency round while that keeps the chunks short (strips, small grids); on large grids
    // several rounds of 14-row chunks beat one round of long ones (4096^2: 184 vs 195 us, 8192^2:
    // 665 vs 758 us) -- the halo rows of adjacent, simultaneously resident chunks are L2 hits
    int R = h->mom_rows > 0 ? h->mom_rows : chunk_rows_fit(h, ntt, resident_waves(h, k_momentum<T, V>), 4, 64);
    if (h->mom_rows <= 0 && R > 32) R = 14;
    const TbPlan tp = tb_plan(h, adapt_par);   // (one extra block: the planner wave)
    launch(h, kMomentum, k_momentum<T, V>, dim3(blocks_for(h, ntt, R) + (tp.masks ? 1u : 0u)), 0, h->g, C(h), (const T*)F_<T>(h, fF),
           (const T*)F_<T>(h, fU), (const T*)F_<T>(h, fV), F_<T>(h, fUS), F_<T>(h, fVS), F_<T>(h, fRHS), R, ntt,
           virt ? 1 : 0, tp);
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
  // the work plan of the step's five-sweep launches (see tb_make_plan): active on parity-keyed step
  // sequences (adapt_par = istep & 1), square or not, two columns per lane, up to TB_COLS tile columns
  static int tb_cols(const vof2d_ctx* h) { return (sizeof(T) == 4 && h->tb_wide) ? 4 : V; }   // columns per lane of the fused Jacobi
  static TbPlan tb_plan(vof2d_ctx* h, int adapt_par) {
    TbPlan tp{nullptr, nullptr, 0, 0, 0, 0};
    if (adapt_par < 0 || !h->tb_adapt || h->tb < 5 || h->tb_narrow == 2 || h->tb_rows > 0) return tp;
    const Consts<T> cc = C(h);
    const bool sq = cc.dxi2 == cc.dyi2 && !h->tb_general;
    int ntt = 0;
    int R;
    if constexpr (sizeof(T) == 4) R = tb_cols(h) == 4 ? jacobi_tb_plan<5, 4>(h, sq, ntt) : jacobi_tb_plan<5, V>(h, sq, ntt);
    else R = jacobi_tb_plan<5, V>(h, sq, ntt);
    const long waves = (long)blocks_for(h, ntt, R) * 4;
    if (ntt > TB_COLS || waves > kTbPlanWaves || (R < 32 && ntt >= 48 && h->tb_narrow != 0)) return tp;
    tp.masks = h->d_tbmask;
    tp.plan = h->d_tbmask + 2 * TB_BANDS * (TB_COLS / 64);
    tp.ntt = ntt; tp.R = R; tp.waves = (int)waves; tp.par = adapt_par;
    return tp;
  }
  template <int TS, int VV>
  static void jacobi_tb_launch(vof2d_ctx* h, const Consts<T>& cc, bool sq, int src, int dst, int R, int ntt, int adapt_par = -1) {
    unsigned long long* none = nullptr;
    TbPlan tp{nullptr, nullptr, 0, 0, 0, 0};
    if (TS == 5 && VV == tb_cols(h)) tp = tb_plan(h, adapt_par);
    if (sq)
      launch(h, kJacobiTB, k_jacobi_tb<T, VV, TS, true, false>, dim3(blocks_for(h, ntt, R)), 0, h->g, cc,
             (const T*)F_<T>(h, src), (const T*)F_<T>(h, fRHS), F_<T>(h, dst), R, ntt, none, tp);
    else
      launch(h, kJacobiTB, k_jacobi_tb<T, VV, TS, false, false>, dim3(blocks_for(h, ntt, R)), 0, h->g, cc,
             (const T*)F_<T>(h, src), (const T*)F_<T>(h, fRHS), F_<T>(h, dst), R, ntt, none, tp);
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
             (const T*)F_<T>(h, src), (const T*)F_<T>(h, fRHS), F_<T>(h, dst), R, ntt, h->d_courant + 1, notp);
    else
      launch(h, kJacobiTB, k_jacobi_tb<T, V, TS, false, true>, dim3(blocks_for(h, ntt, R)), 0, h->g, cc,
             (const T*)F_<T>(h, src), (const T*)F_<T>(h, fRHS), F_<T>(h, dst), R, ntt, h->d_courant + 1, notp);
  }
  template <int TS>
  static void jacobi_tb(vof2d_ctx* h, int src, int dst, int adapt_par = -1) {
    const Consts<T> cc = C(h);
    const bool sq = cc.dxi2 == cc.dyi2 && !h->tb_general;  // square cells: the product-carrying pipeline
    int ntt = 0;
    const int R = jacobi_tb_plan<TS, V>(h, sq, ntt);
    // Thin, wide strips (what strong scaling produces: 1056 x 8192 per GPU at 8 GPUs): the variant
    // with one column per lane needs 71 VGPRs, so 7 waves/SIMD are resident instead of 3.  It wins
    // only while the tiny-value front of the pressure iteration crosses the strip (steps ~65-800 of
    // a run started from p = 0: 71 vs 78-82 us per launch, the slow waves are smaller); before and
    // after that two columns per lane are faster (49-53 vs 59-69 us per launch, 384 vs 401 us per
    // strip step), so it is opt-in (jacobi_tb_narrow = 1).  On full grids two columns per lane win
    // throughout (4096^2: 89 vs 103 us, 2048^2: 30 vs 35 us).  Square cells, five sweeps, fp64 only.
    if constexpr (sizeof(T) == 8 && TS == 5 && V == 2) {
      if (sq && h->tb_rows <= 0 && ((R < 32 && ntt >= 48 && h->tb_narrow != 0) || h->tb_narrow == 2)) {   // 2: on any grid
        int ntt1 = 0;
        const int R1 = jacobi_tb_plan<TS, 1>(h, sq, ntt1);
        jacobi_tb_launch<TS, 1>(h, cc, sq, src, dst, R1, ntt1);
        return;
      }
    }
    if constexpr (sizeof(T) == 4) {
      if (h->tb_wide && h->tb_rows <= 0) {   // fp32: 16 bytes per lane
        int ntt4 = 0;
        const int R4 = jacobi_tb_plan<TS, 4>(h, sq, ntt4);
        jacobi_tb_launch<TS, 4>(h, cc, sq, src, dst, R4, ntt4, adapt_par);
        return;
      }
    }
    jacobi_tb_launch<TS, V>(h, cc, sq, src, dst, R, ntt, adapt_par);
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
    launch(h, kTransport, k_transport<T, V, YFIRST>, dim3((unsigned)((range_chunks(rr) * h->nty + 3) / 4)), 0, h->g, C(h),
           (const T*)F_<T>(h, fF), F_<T>(h, fF2), h->nty, (const T*)F_<T>(h, fUS), (const T*)F_<T>(h, fVS),
           (const T*)F_<T>(h, fP), F_<T>(h, fU), F_<T>(h, fV), h->d_courant, rr);
  }
};

void swap_F(vof2d_ctx* h) {
  void* t = h->fld[fF];
  h->fld[fF] = h->fld[fF2];
  h->fld[fF2] = t;
}

template <typename T, bool POST, bool CORR = false>
void sweep_x(vof2d_ctx* h) { L<T>::template fct_x<POST, CORR>(h); swap_F(h); }
template <typename T, bool POST, bool CORR = false>
void sweep_y(vof2d_ctx* h) { L<T>::template fct_y<POST, CORR>(h); swap_F(h); }
// The second sweep of a step produces the final F.  On a strip only the owned rows are produced
// (the halo rows are the neighbours' to send), and they can be produced in two parts: the
// VOF_HALO_ROWS-row bands next to the interior edges (what the neighbours wait for) and the rest.
enum TransportPart { kAllOwned = 0, kEdgeBands = 1, kRest = 2 };
template <typename T>
void final_sweep(vof2d_ctx* h, bool along_x, int part) {
  const int W = VOF_HALO_ROWS(h->d.jacobi_iters);
  const int lo = h->d.own_lo > h->g.ilo ? h->d.own_lo : h->g.ilo, hi = h->d.own_hi < h->g.ihi ? h->d.own_hi : h->g.ihi;
  const bool band_lo = !h->g.wall_lo, band_hi = !h->g.wall_hi;
  auto run = [&](int a, int b, bool bc) {
    if (b < a) return;
    if (along_x) L<T>::template fct_x<true, false>(h, a, b); else L<T>::template fct_y<true, false>(h, a, b);
    if (bc) L<T>::bc_F_cols(h, F_<T>(h, fF2), a, b);  // rows about to be shipped carry their ghost columns
  };
  if (part == kAllOwned) { run(lo, hi, false); return; }
  const int in_lo = band_lo ? lo + W : lo, in_hi = band_hi ? hi - W : hi;  // strips are >= W rows thick
  const bool split = in_lo <= in_hi && (band_lo || band_hi);
  if (part == kEdgeBands) {
    if (!split) { if (band_lo || band_hi) run(lo, hi, true); return; }  // the bands meet: everything is edge
    if (band_lo) run(lo, in_lo - 1, true);
    if (band_hi) run(in_hi + 1, hi, true);
  } else {
    if (split) run(in_lo, in_hi, false);
    else if (!band_lo && !band_hi) run(lo, hi, false);  // a full domain has no bands: the rest is everything
  }
}

// The fused transport (k_transport) on the owned rows of a strip: all at once (kAllOwned), only the
// two W-row bands at its interior edges (kEdgeBands: both in ONE launch), or only the rest (kRest).
template <typename T>
void transport_part(vof2d_ctx* h, bool y_first, int part) {
  const int W = VOF_HALO_ROWS(h->d.jacobi_iters);
  const int lo = h->d.own_lo > h->g.ilo ? h->d.own_lo : h->g.ilo, hi = h->d.own_hi < h->g.ihi ? h->d.own_hi : h->g.ihi;
  const bool band_lo = !h->g.wall_lo, band_hi = !h->g.wall_hi;
  const int in_lo = band_lo ? lo + W : lo, in_hi = band_hi ? hi - W : hi;   // strips are >= W rows thick
  const bool split = in_lo <= in_hi && (band_lo || band_hi);
  // the bands are few rows: short chunks, so that they are many short-lived waves (2 x 16 rows of an
  // 8192-wide strip: 31 us with 16-row chunks, 15-18 us with 4-row chunks)
  const int Rb = h->band_rows, R = L<T>::transport_rows(h);
  RowRanges rr{{1, 1, 1}, {0, 0, 0}, {Rb, Rb, R}};
  if (part == kAllOwned || !split) {
    // one range: a full domain has no bands (everything is "rest"); where the bands meet there is
    // no rest (everything is "bands")
    if (part == kRest && (band_lo || band_hi)) return;
    if (part == kEdgeBands && !(band_lo || band_hi)) return;
    rr.first[2] = lo; rr.last[2] = hi;
  } else if (part == kEdgeBands) {
    if (band_lo) { rr.first[0] = lo; rr.last[0] = in_lo - 1; }
    if (band_hi) { rr.first[1] = in_hi + 1; rr.last[1] = hi; }
  } else {
    rr.first[2] = in_lo; rr.last[2] = in_hi;
  }
  if (y_first) L<T>::template transport<true>(h, &rr); else L<T>::template transport<false>(h, &rr);
}

// interior copy src -> dst (only used to keep p in place for odd sweep counts)
template <typename T>
__global__ void k_copy_interior(Geom g, const T* __restrict__ s, T* __restrict__ d) {
  const int j = 1 + blockIdx.x * blockDim.x + threadIdx.x;
  const int i = g.ilo + blockIdx.y;
  if (j > g.ny || i > g.ihi) return;
  const size_t o = at(g, i, j);
  d[o] = s[o];
}

template <typename T>
void copy_interior(vof2d_ctx* h, int src, int dst) {
  dim3 grid((h->g.ny + 255) / 256, h->g.ihi - h->g.ilo + 1);
  hipLaunchKernelGGL(k_copy_interior<T>, grid, dim3(256), 0, h->stream, h->g, F_<T>(h, src), F_<T>(h, dst));
}

// n Jacobi sweeps starting from fld[fP]; the result ends in fld[fP] (no pointer swap, so p's ghost
// cells keep their set_BC values like the reference's copy-back loop :265-266).  Sweeps are grouped
// into launches of h->tb fused sweeps (k_jacobi_tb); the remainder and the residual variant use the
// single-sweep kernel.
template <typename T>
void jacobi_n(vof2d_ctx* h, int n, bool resid_last, int adapt_par = -1) {
  if (n <= 0) return;
  if (!(h->tb_adapt && h->tb >= 5 && !resid_last)) adapt_par = -1;
  int cur = fP, oth = fPT;
  auto flip = [&]() { int t = cur; cur = oth; oth = t; };
  int left = n;
  const int tb = h->tb;
  // the last launch carries the norm reductions of its last sweep: a fused launch where the sweep
  // count and the handle's fusion depth allow one, else the single-sweep kernel
  const int last = !resid_last ? 0 : ((tb >= 5 && n >= 5) ? 5 : ((tb >= 2 && n >= 2) ? 2 : 1));
  left -= last;
  while (left > 0) {
    if (tb >= 5 && left >= 5) { L<T>::template jacobi_tb<5>(h, cur, oth, adapt_par); left -= 5; }
    else if (tb >= 2 && left >= 2) { L<T>::template jacobi_tb<2>(h, cur, oth); left -= 2; }
    else { L<T>::template jacobi<false>(h, cur, oth); left -= 1; }
    flip();
  }
  if (last == 5) { L<T>::template jacobi_tb_resid<5>(h, cur, oth); flip(); }
  else if (last == 2) { L<T>::template jacobi_tb_resid<2>(h, cur, oth); flip(); }
  else if (last == 1) { L<T>::template jacobi<true>(h, cur, oth); flip(); }
  if (cur != fP) copy_interior<T>(h, fPT, fP);
}

// The fused per-step schedule, 2dvof.py:506-528 (DESIGN.md "schedule"), in three phases so a
// multi-GPU driver can ship each field's halo as soon as the field is final for the step:
//   phase 0: predictor + pressure solve                      -> p final
//   phase 1: velocity correction + first FCT sweep + BC(u,v) -> u, v final
//   phase 2: second FCT sweep (+post_process_f) + BC(F)      -> F final
// update_uv (:524) is folded into whichever FCT sweep runs first (that sweep streams F anyway and
// needs the corrected velocity): p, F, u*, v* -> u, v does not cost its own 6-pass kernel.
// The reference applies the full set_BC three times per step (:518, :525, :528).  Here each field
// gets its boundary condition once, as soon as it is final for the step -- p (and F, whose ghosts
// the sweeps read; only the first step changes them) after the Jacobi sweeps, u / v after the
// correction, F after the transport:
//   * :518 only rewrites ghosts that :525 rewrites again before anything reads them (p ghosts are
//     read by the Jacobi stencil, but always multiplied by a zero coefficient);
//   * u, v, p do not change after :525, so :528 rewrites identical values for them;
//   * the first sweep derives the boundary values of u, v it needs itself (corrected_velocity), and
//     writes them where the second sweep reads them.
// After every phase the ghost cells of the fields final so far hold exactly what the reference's
// calls leave there, and an in-flight halo receive of a field never overlaps a kernel that writes
// the same field.  vof_step on one handle is the three phases back to back; with merge_bc the
// u, v boundary condition moves behind the second sweep and shares F's launch (full domains only:
// a strip driver wants u, v complete before it ships them).
// lean: no boundary launch inside the phases -- the caller applies set_bc<u,v,F,p> once, after the
// second sweep (and after the halo exchange of a strip).  Valid on a step that starts with F's
// ghost cells already consistent (every step but the first after set_init_F / from_numpy / a
// single verb): p's ghosts only ever feed values the wall conditions override (u[1] = 0, v[:,1] =
// 0) or zero stencil coefficients, and the first sweep itself stores the wall-face zeros of u, v
// the second sweep reads.
template <typename T>
void enqueue_phase(vof2d_ctx* h, int phase, int64_t istep, bool merge_bc = false, bool lean = false, bool virt = false,
                   int adapt_par = -1 /* istep & 1 when the caller's launch sequence is keyed by the step parity */) {
  const bool y_first = (istep % 2 == 0);    // :526, :312-318
  const bool corr = h->fuse_correct != 0;
  if (lean && !(corr && h->fuse_momentum)) lean = false;
  if (phase == 0) {
    // cal_nu_rho (:513) is folded into its consumers: rho/nu = f(F[i,j]) recomputed per cell
    if (h->fuse_momentum) {
      // :514, :517 and the (sweep-invariant, BC-independent) rhs of :239-241 in one pass
      L<T>::momentum(h, virt, adapt_par);   // virt: the previous step's set_BC launch was left out (see enqueue_step)
    } else {
      L<T>::normals(h);                     // :514 loop 1
      L<T>::kappa(h);                       // :514 loop 2
      L<T>::template predictor<false>(h);   // :517
      L<T>::template rhs<false>(h);         // :521-522, rhs part (iteration invariant)
    }
    jacobi_n<T>(h, h->d.jacobi_iters, false, h->fuse_momentum ? adapt_par : -1);  // :521-522
    if (!lean) L<T>::template set_bc<BC_P | BC_F>(h);  // p part of :525 / :528; F part of :518 (first step)
  } else if (phase == 1) {
    if (corr) {                             // :524 inside the first sweep of :526
      if (y_first) sweep_y<T, false, true>(h); else sweep_x<T, false, true>(h);
    } else {
      L<T>::template correct<false>(h);     // :524
    }
    if (!(merge_bc && corr) && !lean) L<T>::template set_bc<BC_UV>(h);  // u, v part of :525
    if (!corr) { if (y_first) sweep_y<T, false>(h); else sweep_x<T, false>(h); }
  } else {
    // second sweep, :527 fused: phase 2 = all owned rows; 3 = the edge bands only (then F's halo
    // rows can leave while) 4 = the remaining rows (are produced); both write the same buffer
    const int part = phase == 2 ? kAllOwned : (phase == 3 ? kEdgeBands : kRest);
    final_sweep<T>(h, /*along_x=*/y_first, part);
    if (phase == 3) return;                 // the new F stays in the twin buffer until phase 4
    swap_F(h);
    if (lean) return;                       // the caller's single set_bc<u,v,F,p> follows
    // F part of :528 on the rows this handle produced; a strip's halo rows arrive with the
    // sender's ghost columns (and may be arriving right now)
    if (merge_bc && corr) L<T>::template set_bc<BC_UV | BC_F>(h);
    else L<T>::template set_bc<BC_F>(h, /*own_rows_only=*/true);
  }
}
template <typename T>
void enqueue_step(vof2d_ctx* h, int64_t istep, bool lean = false, bool virt = false) {
  const bool full = h->g.wall_lo && h->g.wall_hi;
  lean = lean && h->fuse_correct && h->fuse_momentum;
  if (lean && full && h->fuse_transport) {
    // :524 + :526-527 as ONE kernel: the first sweep's F never goes to memory.  One swap of the
    // F / twin pair per step (the two-kernel form swaps twice).
    // With virtual ghosts the step's one set_BC launch goes as well: after it, the only reader of
    // ghost cells is the next step's k_momentum (the sweeps meet F's ghosts only at faces whose
    // wall velocity is zero, update_uv overwrites what p's ghosts would enter, the Jacobi stencil
    // multiplies them by zero coefficients), and that kernel forms them from the interior cells
    // itself.  Whoever else looks at the fields goes through settle_ghosts first.
    L<T>::momentum(h, virt, (int)(istep & 1));
    jacobi_n<T>(h, h->d.jacobi_iters, false, (int)(istep & 1));
    if (istep % 2 == 0) L<T>::template transport<true>(h); else L<T>::template transport<false>(h);
    swap_F(h);
    if (!virt) L<T>::template set_bc<BC_ALL>(h);
    return;
  }
  for (int ph = 0; ph < 3; ++ph) enqueue_phase<T>(h, ph, istep, full, lean, false, (int)(istep & 1));
  if (lean) L<T>::template set_bc<BC_ALL>(h);   // :518, :525, :528 in one launch
}

int ensure_ok(vof2d_ctx* h) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(h->err, sizeof(h->err), "kernel launch failed: %s", hipGetErrorString(e));
    return VOF_EHIP;
  }
  return VOF_OK;
}

#define DISPATCH_T(h, expr_d, expr_f) \
  do { if ((h)->d.dtype == VOF_F64) { expr_d; } else { expr_f; } } while (0)

// true if the next vof_step runs the fused full-domain schedule (k_momentum, 2 x k_jacobi_tb,
// k_transport) that leaves the ghost cells virtual
bool step_leaves_ghosts_virtual(const vof2d_ctx* h) {
  return h->g.wall_lo && h->g.wall_hi && h->fuse_transport && h->fuse_correct && h->fuse_momentum &&
         h->virtual_ghosts && !h->f_ghosts_dirty && !h->uv_ghosts_dirty;
}
// Every entry point that reads or writes fields other than through the fused step calls this
// first: if the last step skipped its set_BC launch, run it now (u, v, F with its twin, p).
void settle_ghosts(vof2d_ctx* h) {
  if (!h->ghosts_virtual) return;
  DISPATCH_T(h, L<double>::set_bc<BC_ALL>(h), L<float>::set_bc<BC_ALL>(h));
  h->ghosts_virtual = false;
}

int copy_rows_host(vof2d_ctx* h, int id, int g0, int g1, void* host, size_t nbytes, bool to_host) {
  if (g0 < h->d.row_lo || g1 > h->d.row_hi || g1 < g0) return fail(h, VOF_EINVAL, "row range not stored by this handle");
  const size_t width = (size_t)(h->g.ny + 2) * h->esz;
  const size_t rows = (size_t)(g1 - g0 + 1);
  if (nbytes != width * rows) return fail(h, VOF_EINVAL, "buffer size does not match (rows, ny+2) of the field dtype");
  char* dev = reinterpret_cast<char*>(h->fld[id]) + ((size_t)(g0 - h->d.row_lo) * h->g.pitch + h->g.col0) * h->esz;
  const size_t dpitch = (size_t)h->g.pitch * h->esz;
  if (to_host)
    HIPCHK(h, hipMemcpy2DAsync(host, width, dev, dpitch, width, rows, hipMemcpyDeviceToHost, h->stream));
  else
    HIPCHK(h, hipMemcpy2DAsync(dev, dpitch, host, width, width, rows, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return VOF_OK;
}

void destroy_xchg_graphs(vof2d_ctx* h);
void destroy_graphs(vof2d_ctx* h) {
  destroy_xchg_graphs(h);
  for (int k = 0; k < 2; ++k)
    for (int o = 0; o < 2; ++o)
      if (h->gexec[k][o]) { (void)hipGraphExecDestroy(h->gexec[k][o]); h->gexec[k][o] = nullptr; }
  for (int k = 0; k < 9; ++k)
    if (h->gphase[k]) { (void)hipGraphExecDestroy(h->gphase[k]); h->gphase[k] = nullptr; }
}


// ---- RCCL, bound at run time (dlopen): the library has no link-time dependency on it, and a
// process that already carries an RCCL (PyTorch's) shares that copy instead of loading a second.
struct Rccl {
  void* dl = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*GetVersion)(int*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int version = 0;
  char why[256] = "";
};
Rccl* rccl_bind(Rccl& r);
Rccl* rccl() {
  // C++11 magic static: the binding happens once, also when two handles are created on two threads
  static Rccl r;
  static Rccl* const bound = rccl_bind(r);
  return bound;
}
Rccl* rccl_bind(Rccl& r) {
  const char* cands[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  const char* forced = getenv("VOF2D_RCCL");
  void* dl = (forced && *forced) ? dlopen(forced, RTLD_NOW | RTLD_LOCAL)
                                 : dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);  // a copy the process already mapped
  for (size_t k = 0; !dl && k < sizeof(cands) / sizeof(cands[0]); ++k) dl = dlopen(cands[k], RTLD_NOW | RTLD_LOCAL);
  if (!dl) { snprintf(r.why, sizeof(r.why), "librccl.so.1 not found: %s", dlerror()); return nullptr; }
#define SYM(field, name)                                                              \
  do {                                                                                \
    *reinterpret_cast<void**>(&r.field) = dlsym(dl, name);                            \
    if (!r.field) { snprintf(r.why, sizeof(r.why), "RCCL lacks %s", name); dlclose(dl); return nullptr; } \
  } while (0)
  SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
  SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv"); SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd");
  SYM(GetErrorString, "ncclGetErrorString"); SYM(AllReduce, "ncclAllReduce"); SYM(GetVersion, "ncclGetVersion");
#undef SYM
  (void)r.GetVersion(&r.version);
  r.dl = dl;
  return &r;
}

#define NCCLCHK(h, call)                                                                         \
  do {                                                                                           \
    int r_ = (call);                                                                             \
    if (r_ != 0) {                                                                               \
      snprintf((h)->err, sizeof((h)->err), "%s:%d %s -> %s", __FILE__, __LINE__, #call,          \
               rccl()->GetErrorString(r_));                                                      \
      return VOF_EHIP;                                                                           \
    }                                                                                            \
  } while (0)

void comm_teardown(vof2d_ctx* h) {
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->cstream) (void)hipStreamSynchronize(h->cstream);
  destroy_xchg_graphs(h);  // captured send/recv nodes hold the communicator: they go first
  if (h->d_red) { (void)hipFree(h->d_red); h->d_red = nullptr; }
  if (h->comm && rccl()) (void)rccl()->CommDestroy(h->comm);
  h->comm = nullptr;
  if (h->ev_ready) (void)hipEventDestroy(h->ev_ready);
  if (h->ev_done) (void)hipEventDestroy(h->ev_done);
  for (int k = 0; k < 3; ++k) {
    if (h->ev_fork[k]) (void)hipEventDestroy(h->ev_fork[k]);
    h->ev_fork[k] = nullptr;
  }
  destroy_xchg_graphs(h);
  if (h->cstream) (void)hipStreamDestroy(h->cstream);
  h->ev_ready = h->ev_done = nullptr;
  h->cstream = nullptr;
  h->peer_lo = h->peer_hi = -1;
}

// Halo exchange of the fields in `mask` with both neighbours: W = VOF_HALO_ROWS owned rows out, W
// halo rows in, per side -- a row is `pitch` contiguous elements, so each message is one contiguous
// block of field memory (no packing).  One RCCL group on the communication stream, ordered after
// everything enqueued on the compute stream so far; the compute stream does not wait (comm_join).
void destroy_xchg_graphs(vof2d_ctx* h) {
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 5; ++b)
      for (int o = 0; o < 2; ++o)
        if (h->gxchg[a][b][o]) { (void)hipGraphExecDestroy(h->gxchg[a][b][o]); h->gxchg[a][b][o] = nullptr; }
}
int comm_post(vof2d_ctx* h, unsigned mask, bool f_in_twin = false, int fork = -1) {
  Rccl* r = rccl();
  const int W = VOF_HALO_ROWS(h->d.jacobi_iters);
  const size_t row_bytes = (size_t)h->g.pitch * h->esz, bytes = (size_t)W * row_bytes;
  hipEvent_t ready = fork >= 0 ? h->ev_fork[fork] : h->ev_ready;
  HIPCHK(h, hipEventRecord(ready, h->stream));
  HIPCHK(h, hipStreamWaitEvent(h->cstream, ready, 0));
  static const int ids[4] = {fF, fU, fV, fP};
  NCCLCHK(h, r->GroupStart());
  // Inside the group no early return: a failing send / recv must still be followed by GroupEnd, or
  // the next (eager) exchange would nest inside the group left open and never be issued.
  int first_err = 0;
  const char* what = "";
  auto note = [&](int rc, const char* call) { if (rc != 0 && first_err == 0) { first_err = rc; what = call; } };
  for (int k = 0; k < 4; ++k) {
    if (!(mask & (1u << k))) continue;
    // between the two transport phases the new F still lives in the twin buffer
    char* base = reinterpret_cast<char*>(h->fld[(k == 0 && f_in_twin) ? fF2 : ids[k]]);
    auto row = [&](int g) { return base + (size_t)(g - h->d.row_lo) * row_bytes; };
    if (h->peer_lo >= 0) {
      note(r->Send(row(h->d.own_lo), bytes, /*ncclInt8*/ 0, h->peer_lo, h->comm, h->cstream), "ncclSend(lo)");
      note(r->Recv(row(h->d.own_lo - W), bytes, 0, h->peer_lo, h->comm, h->cstream), "ncclRecv(lo)");
    }
    if (h->peer_hi >= 0) {
      note(r->Send(row(h->d.own_hi - W + 1), bytes, 0, h->peer_hi, h->comm, h->cstream), "ncclSend(hi)");
      note(r->Recv(row(h->d.own_hi + 1), bytes, 0, h->peer_hi, h->comm, h->cstream), "ncclRecv(hi)");
    }
  }
  note(r->GroupEnd(), "ncclGroupEnd");
  if (first_err != 0) {
    snprintf(h->err, sizeof(h->err), "halo exchange: %s -> %s", what, r->GetErrorString(first_err));
    return VOF_EHIP;
  }
  return VOF_OK;
}
// the compute stream waits for every exchange posted so far
int comm_join(vof2d_ctx* h) {
  HIPCHK(h, hipEventRecord(h->ev_done, h->cstream));
  HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_done, 0));
  return VOF_OK;
}

}  // namespace

// =============================================================== C ABI
extern "C" {

int vof_desc_default(vof2d_desc* d, int32_t nx, int32_t ny, int32_t dtype) {
  if (!d || nx < 3 || ny < 3 || (dtype != VOF_F64 && dtype != VOF_F32)) return VOF_EINVAL;
  memset(d, 0, sizeof(*d));
  d->abi_version = VOF_ABI_VERSION;
  d->nx = nx; d->ny = ny; d->dtype = dtype; d->coord_cast_f32 = 1;
  d->row_lo = 0; d->row_hi = nx + 1; d->own_lo = 1; d->own_hi = nx;
  d->jacobi_iters = 10; d->device = -1; d->flags = 0;
  // 2dvof.py:22-33
  d->Lx = 0.1; d->Ly = 0.1; d->rho_l = 1000.0; d->rho_g = 50.0; d->nu_l = 1.0e-6; d->nu_g = 1.5e-5;
  d->sigma = 0.007; d->gx = 0; d->gy = -5; d->dt = 4e-6;
  return VOF_OK;
}

int vof_create(const vof2d_desc* d, void* stream, vof2d_handle* out) {
  if (!d || !out || d->abi_version != VOF_ABI_VERSION) return VOF_EINVAL;
  if (d->nx < 3 || d->ny < 3 || d->row_lo < 0 || d->row_hi > d->nx + 1 || d->row_hi - d->row_lo < 2) return VOF_EINVAL;
  if (d->dtype != VOF_F64 && d->dtype != VOF_F32) return VOF_EINVAL;
  if (d->jacobi_iters < 0) return VOF_EINVAL;
  vof2d_ctx* h = new (std::nothrow) vof2d_ctx();
  if (!h) return VOF_ENOMEM;
  h->err[0] = 0;
  h->d = *d;
  compute_consts(*d, h->cd);
  if (!(d->dtype == VOF_F64 ? divisors_ok<double>(h->cd) : divisors_ok<float>(h->cd))) {
    delete h;
    return VOF_EINVAL;  // a grid/time-step constant with an all-ones significand (see divisors_ok)
  }
  h->esz = d->dtype == VOF_F64 ? 8 : 4;
  h->V = d->dtype == VOF_F64 ? VecWidth<double>::V : VecWidth<float>::V;
  const int W = 64 * h->V;
  Geom& g = h->g;
  g.nx = d->nx; g.ny = d->ny; g.row_lo = d->row_lo; g.row_hi = d->row_hi;
  g.ilo = d->row_lo + 1 > 1 ? d->row_lo + 1 : 1;
  g.ihi = d->row_hi - 1 < d->nx ? d->row_hi - 1 : d->nx;
  g.own_lo = d->own_lo; g.own_hi = d->own_hi;
  g.wall_lo = d->row_lo == 0; g.wall_hi = d->row_hi == d->nx + 1;
  g.ntj = (d->ny + W - 1) / W;
  h->nty = (d->ny + (W - 8) - 1) / (W - 8);
  const int align = 128 / (int)h->esz;  // elements per 128 bytes
  g.col0 = align - 1;                   // j = 1 lands on a 128-byte boundary
  // furthest column any lane touches: the overlapped tiles of k_fct_y / k_jacobi_tb start at most
  // H <= 12 columns left of j = 1 and their last tile may run a full tile past ny.
  long maxcol = (long)d->ny + W + 16;
  g.pitch = ((g.col0 + maxcol + 1 + align - 1) / align) * align;
  const size_t nrows = (size_t)(d->row_hi - d->row_lo + 1);
  h->field_elems = nrows * (size_t)g.pitch + (size_t)align;  // + one 128-byte tail pad
  const char* ev = getenv("VOF2D_ROWS");
  h->rows_override = ev ? atoi(ev) : 0;
  if ((ev = getenv("VOF2D_TB"))) h->tb = atoi(ev);
  if ((ev = getenv("VOF2D_TB_ROWS"))) h->tb_rows = atoi(ev);
  if ((ev = getenv("VOF2D_FCTX_ROWS"))) h->fctx_rows = atoi(ev);
  if ((ev = getenv("VOF2D_TB_GENERAL"))) h->tb_general = atoi(ev);
  if ((ev = getenv("VOF2D_TB_NARROW"))) h->tb_narrow = atoi(ev);
  if ((ev = getenv("VOF2D_FUSE_CORRECT"))) h->fuse_correct = atoi(ev);
  if ((ev = getenv("VOF2D_FUSE_TRANSPORT"))) h->fuse_transport = atoi(ev);
  if ((ev = getenv("VOF2D_VIRTUAL_GHOSTS"))) h->virtual_ghosts = atoi(ev);
  if ((ev = getenv("VOF2D_BAND_ROWS")) && atoi(ev) > 0) h->band_rows = atoi(ev);
  if ((ev = getenv("VOF2D_FUSE_MOMENTUM"))) h->fuse_momentum = atoi(ev);
  if ((ev = getenv("VOF2D_MOM_ROWS"))) h->mom_rows = atoi(ev);

  int rc = VOF_OK;
  do {
    if (d->device >= 0) {
      if (hipSetDevice(d->device) != hipSuccess) { rc = VOF_EHIP; break; }
    }
    if (hipGetDevice(&h->device) != hipSuccess) { rc = VOF_EHIP; break; }
    if (stream) {
      h->stream = reinterpret_cast<hipStream_t>(stream);
    } else {
      if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { rc = VOF_EHIP; break; }
      h->own_stream = true;
    }
    const size_t bytes = h->field_elems * h->esz * NFIELDS;
    if (hipMalloc(reinterpret_cast<void**>(&h->arena), bytes) != hipSuccess) { rc = VOF_ENOMEM; break; }
    if (hipMemsetAsync(h->arena, 0, bytes, h->stream) != hipSuccess) { rc = VOF_EHIP; break; }
    for (int k = 0; k < NFIELDS; ++k) h->fld[k] = h->arena + (size_t)k * h->field_elems * h->esz;
    h->f_home = h->fld[fF];
    if (hipMalloc(reinterpret_cast<void**>(&h->d_courant), 4 * sizeof(unsigned long long)) != hipSuccess) { rc = VOF_ENOMEM; break; }
    if (hipMemsetAsync(h->d_courant, 0, 4 * sizeof(unsigned long long), h->stream) != hipSuccess) { rc = VOF_EHIP; break; }
    if (hipMalloc(reinterpret_cast<void**>(&h->d_tbmask), (2 * TB_BANDS * (TB_COLS / 64) + 1 + kTbPlanWaves) * sizeof(unsigned long long)) != hipSuccess) { rc = VOF_ENOMEM; break; }
    if (hipMemsetAsync(h->d_tbmask, 0, (2 * TB_BANDS * (TB_COLS / 64) + 1 + kTbPlanWaves) * sizeof(unsigned long long), h->stream) != hipSuccess) { rc = VOF_EHIP; break; }
    if (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) { rc = VOF_EHIP; break; }
    if (hipStreamSynchronize(h->stream) != hipSuccess) { rc = VOF_EHIP; break; }
  } while (0);
  if (rc != VOF_OK) {
    (void)hipGetLastError();
    vof_destroy(h);
    return rc;
  }
  *out = h;
  return VOF_OK;
}

int vof_destroy(vof2d_handle h) {
  if (!h) return VOF_EINVAL;
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  destroy_graphs(h);
  for (int k = 0; k < 2 * vof2d_ctx::kMaxTimed; ++k)
    if (h->tev[k]) (void)hipEventDestroy(h->tev[k]);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  comm_teardown(h);
  if (h->vis) (void)hipFree(h->vis);
  if (h->d_courant) (void)hipFree(h->d_courant);
  if (h->d_tbmask) (void)hipFree(h->d_tbmask);
  if (h->arena) (void)hipFree(h->arena);
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return VOF_OK;
}

int vof_set_init_F(vof2d_handle h, int32_t ic) {
  if (!h) return VOF_EINVAL;
  if (ic < 1 || ic > 3) return fail(h, VOF_EINVAL, "ic must be 1, 2 or 3 (2dvof.py:13)");
  settle_ghosts(h);
  DISPATCH_T(h, L<double>::init_F(h, ic), L<float>::init_F(h, ic));
  h->f_ghosts_dirty = true;
  return ensure_ok(h);
}
int vof_set_BC(vof2d_handle h) {
  if (!h) return VOF_EINVAL;
  DISPATCH_T(h, (L<double>::set_bc<BC_ALL | BC_RHO>(h)), (L<float>::set_bc<BC_ALL | BC_RHO>(h)));
  h->f_ghosts_dirty = false;
  h->uv_ghosts_dirty = false;
  h->ghosts_virtual = false;   // this launch is the one a fused step left out
  return ensure_ok(h);
}
int vof_cal_nu_rho(vof2d_handle h) {
  if (!h) return VOF_EINVAL;
  settle_ghosts(h);
  DISPATCH_T(h, L<double>::nu_rho(h), L<float>::nu_rho(h));
  return ensure_ok(h);
}
int vof_get_normal_young(vof2d_handle h) {
  if (!h) return VOF_EINVAL;
  settle_ghosts(h);
  DISPATCH_T(h, (L<double>::normals(h), L<double>::kappa(h)), (L<float>::normals(h), L<float>::kappa(h)));
  return ensure_ok(h);
}
int vof_advect_upwind(vof2d_handle h) {
  if (!h) return VOF_EINVAL;
  settle_ghosts(h);
  DISPATCH_T(h, L<double>::predictor<true>(h), L<float>::predictor<true>(h));
  return ensure_ok(h);
}
int vof_solve_p_jacobi(vof2d_handle h, int32_t n) {
  if (!h) return VOF_EINVAL;
  settle_ghosts(h);
  if (n < 0) return fail(h, VOF_EINVAL, "n must be >= 0");
  if (n == 0) return VOF_OK;
  DISPATCH_T(h, (L<double>::rhs<true>(h), jacobi_n<double>(h, n, false)),
             (L<float>::rhs<true>(h), jacobi_n<float>(h, n, false)));
  return ensure_ok(h);
}
int vof_update_uv(vof2d_handle h) {
  if (!h) return VOF_EINVAL;
  settle_ghosts(h);
  DISPATCH_T(h, L<double>::correct<true>(h), L<float>::correct<true>(h));
  h->uv_ghosts_dirty = true;
  return ensure_ok(h);
}
// A single sweep swaps F with its twin, so field pointers baked into captured step graphs go
// stale: drop the graphs (they are re-captured on the next vof_step / vof_step_phase).
static void sweep_swapped(vof2d_handle h) {
  bool any = h->gexec[0][0] || h->gexec[0][1] || h->gexec[1][0] || h->gexec[1][1];
  for (int k = 0; k < 9; ++k) any = any || h->gphase[k];
  for (int k = 0; k < 20; ++k) any = any || h->gxchg[k / 10][(k / 2) % 5][k % 2];
  if (!any) return;
  (void)hipStreamSynchronize(h->stream);
  destroy_graphs(h);
}
int vof_fct_x_sweep(vof2d_handle h) {
  if (!h) return VOF_EINVAL;
  settle_ghosts(h);
  DISPATCH_T(h, (sweep_x<double, false, false>(h)), (sweep_x<float, false, false>(h)));
  h->f_ghosts_dirty = true;
  sweep_swapped(h);
  return ensure_ok(h);
}
int vof_fct_y_sweep(vof2d_handle h) {
  if (!h) return VOF_EINVAL;
  settle_ghosts(h);
  DISPATCH_T(h, (sweep_y<double, false, false>(h)), (sweep_y<float, false, false>(h)));
  h->f_ghosts_dirty = true;
  sweep_swapped(h);
  return ensure_ok(h);
}
int vof_solve_VOF_rudman(vof2d_handle h, int64_t istep) {
  if (!h) return VOF_EINVAL;
  int rc;
  if (istep % 2 == 0) {
    if ((rc = vof_fct_y_sweep(h))) return rc;
    return vof_fct_x_sweep(h);
  }
  if ((rc = vof_fct_x_sweep(h))) return rc;
  return vof_fct_y_sweep(h);
}
int vof_post_process_f(vof2d_handle h) {
  if (!h) return VOF_EINVAL;
  settle_ghosts(h);
  DISPATCH_T(h, L<double>::post(h), L<float>::post(h));
  h->f_ghosts_dirty = true;
  return ensure_ok(h);
}

int vof_step(vof2d_handle h, int64_t nsteps) {
  if (!h) return VOF_EINVAL;
  if (nsteps < 0) return fail(h, VOF_EINVAL, "nsteps must be >= 0");
  if (h->next_phase != 0) return fail(h, VOF_ESTATE, "a phased step (vof_step_phase) is in progress");
  const bool use_graph = !(h->d.flags & VOF_FLAG_NO_GRAPH);
  for (int64_t s = 0; s < nsteps; ++s) {
    h->istep += 1;
    const int par = (int)(h->istep & 1);
    // A step that starts with consistent F ghosts runs the lean schedule from a captured graph
    // (full domains: k_momentum, 2 x k_jacobi_tb, k_transport and no boundary launch -- virtual
    // ghosts; strips: the two-kernel transport and one boundary launch at the end); the first step
    // after set_init_F / from_numpy / a single verb runs the schedule with the reference's
    // intermediate set_BC calls, eagerly.
    const bool lean = !h->f_ghosts_dirty;
    const bool virt = step_leaves_ghosts_virtual(h);
    if (!virt) settle_ghosts(h);
    // a captured step holds the kernels of the handle's regular schedule; the one step after u / v
    // were written without a set_BC (stored ghost cells must be read as they are) runs eagerly
    const bool regular = !h->uv_ghosts_dirty;
    if (use_graph && lean && regular) {
      // graphs bake the field pointers in: one per (parity, which buffer of the F / twin pair holds
      // F).  The two-kernel transport swaps the pair twice per step, the fused one once.
      const int ori = h->fld[fF] == h->f_home ? 0 : 1;
      const bool one_swap = h->g.wall_lo && h->g.wall_hi && h->fuse_transport && h->fuse_correct && h->fuse_momentum;
      if (!h->gexec[par][ori]) {
        hipGraph_t graph = nullptr;
        HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        DISPATCH_T(h, enqueue_step<double>(h, h->istep, true, virt), enqueue_step<float>(h, h->istep, true, virt));
        HIPCHK(h, hipStreamEndCapture(h->stream, &graph));
        hipError_t e = hipGraphInstantiate(&h->gexec[par][ori], graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) {
          snprintf(h->err, sizeof(h->err), "hipGraphInstantiate: %s", hipGetErrorString(e));
          return VOF_EHIP;
        }
        if (one_swap) swap_F(h);   // capturing ran enqueue_step, which swapped the host's view: undo, redo below
      }
      HIPCHK(h, hipGraphLaunch(h->gexec[par][ori], h->stream));
      if (one_swap) swap_F(h);     // keep the host's view in step with what the replayed kernels did
    } else {
      DISPATCH_T(h, enqueue_step<double>(h, h->istep, lean, virt), enqueue_step<float>(h, h->istep, lean, virt));
      int rc = ensure_ok(h);
      if (rc) return rc;
    }
    h->f_ghosts_dirty = false;
    h->uv_ghosts_dirty = false;
    h->ghosts_virtual = virt;
  }
  return VOF_OK;
}
// The phase graphs bake the F / twin pointers in and assume the pair returns to the same orientation
// after every step (two swaps).  The fused transport swaps once per step, so a handle that mixes
// the entry points may arrive here with the pair the other way round: drop those graphs then
// (they are re-captured on use).  The step and exchange graphs are keyed by the orientation.
static int match_phase_graph_orientation(vof2d_handle h) {
  const int ori = h->fld[fF] == h->f_home ? 0 : 1;
  if (ori == h->phase_graph_ori) return VOF_OK;
  bool any = false;
  for (int k = 0; k < 9; ++k) any = any || h->gphase[k];
  if (any) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int k = 0; k < 9; ++k)
      if (h->gphase[k]) { (void)hipGraphExecDestroy(h->gphase[k]); h->gphase[k] = nullptr; }
  }
  h->phase_graph_ori = ori;
  return VOF_OK;
}
int vof_step_phase(vof2d_handle h, int32_t phase) {
  if (!h) return VOF_EINVAL;
  if (phase < 0 || phase > 4) return fail(h, VOF_EINVAL, "phase must be 0 ... 4");
  // order: 0, 1, then 2 or (3, 4)
  const bool ok = phase == h->next_phase || (phase == VOF_PHASE_TRANSPORT_EDGES && h->next_phase == 2);
  if (!ok) return fail(h, VOF_ESTATE, "vof_step_phase must be called in the order 0, 1, 2 or 0, 1, 3, 4");
  if (phase == 0) {
    settle_ghosts(h);
    int rc = match_phase_graph_orientation(h);
    if (rc) return rc;
    h->istep += 1;
  }
  h->next_phase = (phase == 2 || phase == 4) ? 0 : phase + 1;
  if (phase == 2 || phase == 4) h->f_ghosts_dirty = h->uv_ghosts_dirty = false;   // the phases carry every set_BC of the step
  const bool use_graph = !(h->d.flags & VOF_FLAG_NO_GRAPH);
  if (!use_graph) {
    DISPATCH_T(h, enqueue_phase<double>(h, phase, h->istep), enqueue_phase<float>(h, phase, h->istep));
    return ensure_ok(h);
  }
  const int slot = phase == 0 ? 0 : 2 * phase - 1 + (int)(h->istep & 1);
  bool captured_now = false;
  if (!h->gphase[slot]) {
    hipGraph_t graph = nullptr;
    HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    DISPATCH_T(h, enqueue_phase<double>(h, phase, h->istep), enqueue_phase<float>(h, phase, h->istep));
    HIPCHK(h, hipStreamEndCapture(h->stream, &graph));
    hipError_t e = hipGraphInstantiate(&h->gphase[slot], graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
      snprintf(h->err, sizeof(h->err), "hipGraphInstantiate: %s", hipGetErrorString(e));
      return VOF_EHIP;
    }
    captured_now = true;
  }
  HIPCHK(h, hipGraphLaunch(h->gphase[slot], h->stream));
  // keep the host's view of the F / twin buffers in step with what the replayed kernels did
  // (capturing ran enqueue_phase, which swapped them itself)
  if (!captured_now && (phase == 1 || phase == 2 || phase == 4)) swap_F(h);
  return VOF_OK;
}
int vof_get_istep(vof2d_handle h, int64_t* istep) {
  if (!h || !istep) return VOF_EINVAL;
  *istep = h->istep;
  return VOF_OK;
}
int vof_set_istep(vof2d_handle h, int64_t istep) {
  if (!h) return VOF_EINVAL;
  h->istep = istep;
  return VOF_OK;
}

double vof_residual_value(double max_update, double max_p, int32_t criterion) {
  if (!(max_update < HUGE_VAL)) return HUGE_VAL;   /* inf or NaN: diverged */
  if (criterion == VOF_RESID_ABS) return max_update;
  /* a finite update over a tiny (or zero) max|p_new| must not read as "diverged": the quotient is
   * clamped to the largest finite double, so only a non-finite UPDATE ever returns +inf */
  const double q = max_update / (max_p > VOF_RESID_TINY ? max_p : VOF_RESID_TINY);
  return q < HUGE_VAL ? q : DBL_MAX;
}
int vof_jacobi_sweeps_norms(vof2d_handle h, int32_t n, int32_t build_rhs, double* max_update, double* max_p) {
  if (!h || !max_update || !max_p) return VOF_EINVAL;
  if (n < 1) return fail(h, VOF_EINVAL, "n must be >= 1");
  settle_ghosts(h);
  HIPCHK(h, hipMemsetAsync(h->d_courant + 1, 0, 2 * sizeof(unsigned long long), h->stream));
  if (build_rhs) DISPATCH_T(h, L<double>::rhs<false>(h), L<float>::rhs<false>(h));
  DISPATCH_T(h, jacobi_n<double>(h, n, true), jacobi_n<float>(h, n, true));
  int rc = ensure_ok(h);
  if (rc) return rc;
  unsigned long long bits[2] = {0, 0};
  HIPCHK(h, hipMemcpyAsync(bits, h->d_courant + 1, sizeof(bits), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  memcpy(max_update, &bits[0], sizeof(double));
  memcpy(max_p, &bits[1], sizeof(double));
  return VOF_OK;
}
int vof_jacobi_sweeps_residual(vof2d_handle h, int32_t n, int32_t build_rhs, double* residual) {
  double pmax = 0.0;
  if (!residual) return VOF_EINVAL;
  return vof_jacobi_sweeps_norms(h, n, build_rhs, residual, &pmax);
}

int vof_solve_p(vof2d_handle h, double tol, int32_t max_iters, int32_t check_every, int32_t criterion,
                int32_t* iters_done, double* residual) {
  if (!h || !iters_done || !residual) return VOF_EINVAL;
  if (max_iters < 1 || check_every < 1) return fail(h, VOF_EINVAL, "max_iters and check_every must be >= 1");
  if (criterion != VOF_RESID_ABS && criterion != VOF_RESID_REL) return fail(h, VOF_EINVAL, "criterion must be VOF_RESID_ABS or VOF_RESID_REL");
  int done = 0;
  double r = 0.0;
  bool first = true;
  while (done < max_iters) {
    const int n = check_every < max_iters - done ? check_every : max_iters - done;
    double upd = 0.0, pmax = 0.0;
    int rc = vof_jacobi_sweeps_norms(h, n, first ? 1 : 0, &upd, &pmax);
    if (rc) return rc;
    first = false;
    done += n;
    r = vof_residual_value(upd, pmax, criterion);
    if (r <= tol || !(r < HUGE_VAL)) break;   // converged, or diverged (a non-finite update reads +inf)
  }
  *iters_done = done;
  *residual = r;
  return VOF_OK;
}
int vof_solve_p_residual(vof2d_handle h, double tol, int32_t max_iters, int32_t check_every, int32_t* iters_done,
                         double* residual) {
  return vof_solve_p(h, tol, max_iters, check_every, VOF_RESID_ABS, iters_done, residual);
}

int vof_get_rows(vof2d_handle h, const char* name, int32_t g0, int32_t g1, void* dst, size_t nbytes) {
  if (!h || !dst) return VOF_EINVAL;
  settle_ghosts(h);
  int id = field_id(name);
  if (id < 0) return fail(h, VOF_EINVAL, "unknown field name");
  return copy_rows_host(h, id, g0, g1, dst, nbytes, true);
}
int vof_set_rows(vof2d_handle h, const char* name, int32_t g0, int32_t g1, const void* src, size_t nbytes) {
  if (!h || !src) return VOF_EINVAL;
  settle_ghosts(h);
  int id = field_id(name);
  if (id < 0) return fail(h, VOF_EINVAL, "unknown field name");
  int rc = copy_rows_host(h, id, g0, g1, const_cast<void*>(src), nbytes, false);
  if (rc == VOF_OK && id == fF) rc = copy_rows_host(h, fF2, g0, g1, const_cast<void*>(src), nbytes, false);
  if (id == fF || id == fF2) h->f_ghosts_dirty = true;
  if (id == fU || id == fV) h->uv_ghosts_dirty = true;
  return rc;
}
int vof_get_field(vof2d_handle h, const char* name, void* dst, size_t nbytes) {
  if (!h) return VOF_EINVAL;
  return vof_get_rows(h, name, h->d.row_lo, h->d.row_hi, dst, nbytes);
}
int vof_set_field(vof2d_handle h, const char* name, const void* src, size_t nbytes) {
  if (!h) return VOF_EINVAL;
  return vof_set_rows(h, name, h->d.row_lo, h->d.row_hi, src, nbytes);
}
int vof_field_view(vof2d_handle h, const char* name, void** base, int64_t* pitch, int64_t* col0, int64_t* nrows) {
  if (!h) return VOF_EINVAL;
  settle_ghosts(h);
  int id = field_id(name);
  if (id < 0) return fail(h, VOF_EINVAL, "unknown field name");
  if (base) *base = h->fld[id];
  if (pitch) *pitch = h->g.pitch;
  if (col0) *col0 = h->g.col0;
  if (nrows) *nrows = h->d.row_hi - h->d.row_lo + 1;
  return VOF_OK;
}
int vof_copy_rows(vof2d_handle dst, vof2d_handle src, const char* name, int32_t g0, int32_t g1) {
  if (!dst || !src) return VOF_EINVAL;
  settle_ghosts(dst);
  settle_ghosts(src);
  int id = field_id(name);
  if (id < 0) return fail(dst, VOF_EINVAL, "unknown field name");
  if (dst->d.ny != src->d.ny || dst->d.nx != src->d.nx || dst->d.dtype != src->d.dtype)
    return fail(dst, VOF_EINVAL, "handles differ in nx, ny or dtype");
  if (g1 < g0 || g0 < src->d.row_lo || g1 > src->d.row_hi || g0 < dst->d.row_lo || g1 > dst->d.row_hi)
    return fail(dst, VOF_EINVAL, "rows not stored by both handles");
  // both use the same pitch/col0 (functions of ny and dtype only): one contiguous block
  const size_t off_s = (size_t)(g0 - src->d.row_lo) * src->g.pitch * src->esz;
  const size_t off_d = (size_t)(g0 - dst->d.row_lo) * dst->g.pitch * dst->esz;
  const size_t bytes = (size_t)(g1 - g0 + 1) * src->g.pitch * src->esz;
  // order: after src's pending work, on dst's stream
  HIPCHK(dst, hipEventRecord(src->ev1, src->stream));
  HIPCHK(dst, hipStreamWaitEvent(dst->stream, src->ev1, 0));
  HIPCHK(dst, hipMemcpyAsync(reinterpret_cast<char*>(dst->fld[id]) + off_d,
                             reinterpret_cast<char*>(src->fld[id]) + off_s, bytes, hipMemcpyDeviceToDevice,
                             dst->stream));
  if (id == fF)
    HIPCHK(dst, hipMemcpyAsync(reinterpret_cast<char*>(dst->fld[fF2]) + off_d,
                               reinterpret_cast<char*>(src->fld[fF]) + off_s, bytes, hipMemcpyDeviceToDevice,
                               dst->stream));
  if (dst->g.wall_lo && dst->g.wall_hi) {  // a full domain: the rows' neighbours' ghost cells may no longer mirror them
    if (id == fF) dst->f_ghosts_dirty = true;
    if (id == fU || id == fV) dst->uv_ghosts_dirty = true;
  }
  return VOF_OK;
}

// 2dvof.py:458-492 -- display fields.  The image / vector field is produced on the device into a
// scratch buffer allocated on first use and copied to the caller's dense host array.
static int vis_scratch(vof2d_handle h, size_t bytes) {
  if (h->vis_bytes >= bytes) return VOF_OK;
  if (h->vis) (void)hipFree(h->vis);
  h->vis = nullptr;
  h->vis_bytes = 0;
  if (hipMalloc(&h->vis, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return fail(h, VOF_ENOMEM, "hipMalloc of the visualisation buffer failed");
  }
  h->vis_bytes = bytes;
  return VOF_OK;
}
int vof_get_vis_field(vof2d_handle h, const char* which, void* dst, size_t nbytes) {
  if (!h || !which || !dst) return VOF_EINVAL;
  settle_ghosts(h);
  if (!(h->g.wall_lo && h->g.wall_hi)) return fail(h, VOF_ESTATE, "display fields need a full-domain handle");
  int mode = !strcmp(which, "vof") ? 0 : !strcmp(which, "u") ? 1 : !strcmp(which, "v") ? 2 : !strcmp(which, "vnorm") ? 3 : -1;
  if (mode < 0) return fail(h, VOF_EINVAL, "display field must be vof, u, v or vnorm");
  const size_t bytes = (size_t)4 * h->g.nx * h->g.ny * h->esz;
  if (nbytes != bytes) return fail(h, VOF_EINVAL, "buffer must be (2*nx, 2*ny) of the field dtype");
  int rc = vis_scratch(h, bytes);
  if (rc) return rc;
  dim3 grid((2 * h->g.ny + 255) / 256, 2 * h->g.nx);
  const double umax = h->d.Lx / 0.2, vmax = h->d.Ly / 0.2;  // :468, :476, :484
  if (h->d.dtype == VOF_F64)
    launch(h, kOther, k_vis_field<double>, grid, 0, h->g, (const double*)F_<double>(h, fF), (const double*)F_<double>(h, fU),
           (const double*)F_<double>(h, fV), (double*)h->vis, mode, umax, vmax);
  else
    launch(h, kOther, k_vis_field<float>, grid, 0, h->g, (const float*)F_<float>(h, fF), (const float*)F_<float>(h, fU),
           (const float*)F_<float>(h, fV), (float*)h->vis, mode, (float)umax, (float)vmax);
  HIPCHK(h, hipMemcpyAsync(dst, h->vis, bytes, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return ensure_ok(h);
}
int vof_interp_velocity(vof2d_handle h, void* dst, size_t nbytes) {
  if (!h || !dst) return VOF_EINVAL;
  settle_ghosts(h);
  if (!(h->g.wall_lo && h->g.wall_hi)) return fail(h, VOF_ESTATE, "interp_velocity needs a full-domain handle");
  const size_t bytes = (size_t)2 * (h->g.nx + 2) * (h->g.ny + 2) * h->esz;
  if (nbytes != bytes) return fail(h, VOF_EINVAL, "buffer must be (nx+2, ny+2, 2) of the field dtype");
  int rc = vis_scratch(h, bytes);
  if (rc) return rc;
  dim3 grid((h->g.ny + 2 + 255) / 256, h->g.nx + 2);
  if (h->d.dtype == VOF_F64)
    launch(h, kOther, k_interp_velocity<double>, grid, 0, h->g, (const double*)F_<double>(h, fU),
           (const double*)F_<double>(h, fV), (double*)h->vis);
  else
    launch(h, kOther, k_interp_velocity<float>, grid, 0, h->g, (const float*)F_<float>(h, fU),
           (const float*)F_<float>(h, fV), (float*)h->vis);
  HIPCHK(h, hipMemcpyAsync(dst, h->vis, bytes, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return ensure_ok(h);
}

int vof_set_param(vof2d_handle h, const char* name, double value) {
  if (!h || !name) return VOF_EINVAL;
  if (!strcmp(name, "sigma")) {  // sigma[None] = value (2dvof.py:28-29); constants are baked into graphs
    h->d.sigma = value;
    h->cd.sigma = value;
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    destroy_graphs(h);
    return VOF_OK;
  }
  settle_ghosts(h);
  if (!strcmp(name, "jacobi_tb") || !strcmp(name, "jacobi_tb_rows") || !strcmp(name, "momentum_rows") ||
      !strcmp(name, "fuse_momentum") || !strcmp(name, "fuse_correct") || !strcmp(name, "fuse_transport") ||
      !strcmp(name, "virtual_ghosts") || !strcmp(name, "band_rows") || !strcmp(name, "fctx_rows") ||
      !strcmp(name, "fctx_corr_rows") || !strcmp(name, "jacobi_tb_narrow") || !strcmp(name, "jacobi_tb_adapt") ||
      !strcmp(name, "jacobi_tb_wide")) {  // tuning knobs
    if (!strcmp(name, "jacobi_tb")) h->tb = (int)value;
    else if (!strcmp(name, "jacobi_tb_adapt")) h->tb_adapt = (int)value;
    else if (!strcmp(name, "jacobi_tb_wide")) h->tb_wide = (int)value;
    else if (!strcmp(name, "fctx_rows")) h->fctx_rows = (int)value;
    else if (!strcmp(name, "fctx_corr_rows")) h->fctx_corr_rows = (int)value;
    else if (!strcmp(name, "jacobi_tb_narrow")) h->tb_narrow = (int)value;
    else if (!strcmp(name, "jacobi_tb_rows")) h->tb_rows = (int)value;
    else if (!strcmp(name, "momentum_rows")) h->mom_rows = (int)value;
    else if (!strcmp(name, "fuse_correct")) h->fuse_correct = (int)value;
    else if (!strcmp(name, "fuse_transport")) h->fuse_transport = (int)value;
    else if (!strcmp(name, "virtual_ghosts")) h->virtual_ghosts = (int)value;
    else if (!strcmp(name, "band_rows")) h->band_rows = (int)value < 1 ? 1 : (int)value;
    else h->fuse_momentum = (int)value;
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    destroy_graphs(h);
    return VOF_OK;
  }
  if (!strcmp(name, "rows_per_wave")) {  // tuning knob (0 = heuristic)
    h->rows_override = (int)value;
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    destroy_graphs(h);
    return VOF_OK;
  }
  return fail(h, VOF_EINVAL, "unknown or read-only parameter");
}
int vof_get_param(vof2d_handle h, const char* name, double* value) {
  if (!h || !name || !value) return VOF_EINVAL;
#define P(n) if (!strcmp(name, #n)) { *value = h->cd.n; return VOF_OK; }
  P(sigma) P(dt) P(dx) P(dy) P(dxi) P(dyi) P(dxi2) P(dyi2) P(rho_l) P(rho_g) P(nu_l) P(nu_g) P(gx) P(gy)
  P(nrm_x) P(nrm_y) P(kap_x) P(kap_y) P(dxdy) P(dtdy) P(dtdx) P(cfl_x) P(cfl_y) P(half_dx) P(half_dy)
  P(sqrt2dx) P(tiny)
#undef P
  if (!strcmp(name, "Lx")) { *value = h->d.Lx; return VOF_OK; }
  if (!strcmp(name, "Ly")) { *value = h->d.Ly; return VOF_OK; }
  if (!strcmp(name, "pitch")) { *value = (double)h->g.pitch; return VOF_OK; }
  if (!strcmp(name, "rows_per_wave")) { *value = (double)pick_rows(h, h->g.ntj); return VOF_OK; }
  if (!strcmp(name, "jacobi_tb")) { *value = (double)h->tb; return VOF_OK; }
  if (!strcmp(name, "jacobi_tb_adapt")) { *value = (double)h->tb_adapt; return VOF_OK; }
  if (!strcmp(name, "fuse_transport")) {  // 1 if vof_step runs both FCT sweeps as one kernel on this handle
    *value = (h->g.wall_lo && h->g.wall_hi && h->fuse_transport && h->fuse_correct && h->fuse_momentum) ? 1.0 : 0.0;
    return VOF_OK;
  }
  return fail(h, VOF_EINVAL, "unknown parameter");
}
int vof_get_counter(vof2d_handle h, const char* name, int64_t* value) {
  if (!h || !name || !value) return VOF_EINVAL;
  if (!strcmp(name, "courant_violations")) {
    unsigned long long v = 0;
    HIPCHK(h, hipMemcpyAsync(&v, h->d_courant, sizeof(v), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *value = (int64_t)v;
    return VOF_OK;
  }
  if (!strcmp(name, "tb_plan_active")) {   // 1 if the last fused step's k_jacobi_tb launches ran the equal-cost work plan (tb_make_plan)
    unsigned long long v = 0;
    HIPCHK(h, hipMemcpyAsync(&v, h->d_tbmask + 2 * TB_BANDS * (TB_COLS / 64), sizeof(v), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *value = (int64_t)v;
    return VOF_OK;
  }
  if (!strcmp(name, "exchange_graph_steps")) {  // steps vof_step_exchange replayed from a captured graph
    *value = h->xchg_graph_steps;
    return VOF_OK;
  }
  return fail(h, VOF_EINVAL, "unknown counter");
}

int vof_sync(vof2d_handle h) {
  if (!h) return VOF_EINVAL;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return ensure_ok(h);
}
int vof_timer_start(vof2d_handle h) {
  if (!h) return VOF_EINVAL;
  HIPCHK(h, hipEventRecord(h->ev0, h->stream));
  return VOF_OK;
}
int vof_timer_stop(vof2d_handle h, float* ms) {
  if (!h || !ms) return VOF_EINVAL;
  HIPCHK(h, hipEventRecord(h->ev1, h->stream));
  HIPCHK(h, hipEventSynchronize(h->ev1));
  HIPCHK(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
  return VOF_OK;
}
// nsteps steps of the fused schedule launched eagerly with a start/stop event pair on every
// dispatch; durations accumulate per kernel (vof_get_profile).  Steps are enqueued in batches
// without host synchronisation in between (an idle GPU drops its clocks).
int vof_profile_steps(vof2d_handle h, int64_t nsteps) {
  if (!h) return VOF_EINVAL;
  if (nsteps < 0) return fail(h, VOF_EINVAL, "nsteps must be >= 0");
  if (h->next_phase != 0) return fail(h, VOF_ESTATE, "a phased step (vof_step_phase) is in progress");
  for (int k = 0; k < 2 * vof2d_ctx::kMaxTimed; ++k)
    if (!h->tev[k]) HIPCHK(h, hipEventCreate(&h->tev[k]));
  const int per_step = 16 + h->d.jacobi_iters;  // upper bound of launches in one step
  int64_t done = 0;
  while (done < nsteps) {
    h->timed = 0;
    int batch = 0;
    while (done + batch < nsteps && h->timed + per_step <= vof2d_ctx::kMaxTimed) {
      h->istep += 1;
      const bool lean = !h->f_ghosts_dirty;
      const bool virt = step_leaves_ghosts_virtual(h);
      if (!virt) settle_ghosts(h);
      DISPATCH_T(h, enqueue_step<double>(h, h->istep, lean, virt), enqueue_step<float>(h, h->istep, lean, virt));
      h->f_ghosts_dirty = false;
      h->uv_ghosts_dirty = false;
      h->ghosts_virtual = virt;
      ++batch;
    }
    const int launches = h->timed;
    h->timed = -1;
    if (batch == 0) return fail(h, VOF_ESTATE, "a step has more launches than the profiling event pool");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    // Dispatch start -> stop.  A dispatch's start stamp is taken when the command processor picks
    // the packet up, while the predecessor's last waves are still draining, so for kernels that
    // follow a long-tailed kernel the figure includes that overlap (the per-step sum can exceed the
    // wall time by ~5 %); it is a diagnostic breakdown, rocprofv3 gives exclusive times.
    for (int k = 0; k < launches; ++k) {
      float ms = 0.f;
      HIPCHK(h, hipEventElapsedTime(&ms, h->tev[2 * k], h->tev[2 * k + 1]));
      h->prof_sum_ms[h->tkid[k]] += ms;
      h->prof_cnt[h->tkid[k]] += 1;
    }
    done += batch;
  }
  return ensure_ok(h);
}
int vof_get_profile(vof2d_handle h, const char* kernel, double* avg_us, int64_t* launches) {
  if (!h || !kernel) return VOF_EINVAL;
  for (int k = 0; k < NKERNELS; ++k)
    if (!strcmp(kernel, kKernelNames[k])) {
      if (avg_us) *avg_us = h->prof_cnt[k] ? 1e3 * h->prof_sum_ms[k] / (double)h->prof_cnt[k] : 0.0;
      if (launches) *launches = h->prof_cnt[k];
      return VOF_OK;
    }
  return fail(h, VOF_EINVAL, "unknown kernel name");
}
int vof_reset_profile(vof2d_handle h) {
  if (!h) return VOF_EINVAL;
  for (int k = 0; k < NKERNELS; ++k) { h->prof_sum_ms[k] = 0.0; h->prof_cnt[k] = 0; }
  return VOF_OK;
}
#ifdef VOF_WAVE_TIMES
// Diagnostic build only (make wavetimes; tools/wave_balance.py).  Arms the per-wave start/end
// stamps for kernel `kid` (KernelId) with room for `cap` waves, or reads them back (out != NULL).
extern "C" int vof_debug_wave_times(vof2d_handle h, int32_t kid, uint64_t* out, uint32_t cap) {
  static unsigned long long* buf = nullptr;
  static unsigned int bufcap = 0;
  if (!h) return VOF_EINVAL;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (out) {
    if (!buf || cap > bufcap) return VOF_EINVAL;
    HIPCHK(h, hipMemcpy(out, buf, (size_t)cap * 16, hipMemcpyDeviceToHost));
    return VOF_OK;
  }
  if (cap > bufcap) {
    if (buf) (void)hipFree(buf);
    HIPCHK(h, hipMalloc(&buf, (size_t)cap * 16));
    bufcap = cap;
  }
  HIPCHK(h, hipMemset(buf, 0, (size_t)bufcap * 16));
  int k = kid;
  HIPCHK(h, hipMemcpyToSymbol(HIP_SYMBOL(vof::vof_wave_times), &buf, sizeof(buf)));
  HIPCHK(h, hipMemcpyToSymbol(HIP_SYMBOL(vof::vof_wave_kid), &k, sizeof(k)));
  HIPCHK(h, hipMemcpyToSymbol(HIP_SYMBOL(vof::vof_wave_cap), &bufcap, sizeof(bufcap)));
  return VOF_OK;
}
#endif
int vof_time_jacobi(vof2d_handle h, int32_t n, float* ms_per_sweep) {
  if (!h || !ms_per_sweep) return VOF_EINVAL;
  if (n < 2 || (n & 1)) return fail(h, VOF_EINVAL, "n must be even and >= 2");
  if (h->next_phase != 0) return fail(h, VOF_ESTATE, "a phased step (vof_step_phase) is in progress");
  // one hipEvent pair on the handle's stream around n back-to-back sweeps of the current rhs
  HIPCHK(h, hipEventRecord(h->ev0, h->stream));
  DISPATCH_T(h, jacobi_n<double>(h, n, false), jacobi_n<float>(h, n, false));
  HIPCHK(h, hipEventRecord(h->ev1, h->stream));
  HIPCHK(h, hipEventSynchronize(h->ev1));
  float ms = 0.f;
  HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *ms_per_sweep = ms / (float)n;
  return ensure_ok(h);
}
// ---- strips over RCCL: the per-step halo exchange without leaving the library
int vof_comm_get_unique_id(void* id) {
  if (!id) return VOF_EINVAL;
  Rccl* r = rccl();
  if (!r) return VOF_ESTATE;
  return r->GetUniqueId(id) == 0 ? VOF_OK : VOF_EHIP;
}
int vof_comm_init(vof2d_handle h, const void* id, int32_t rank, int32_t world, int32_t flags) {
  if (!h || !id || world < 1 || rank < 0 || rank >= world) return VOF_EINVAL;
  if (h->comm) return fail(h, VOF_ESTATE, "vof_comm_init: the handle already has a communicator");
  settle_ghosts(h);
  Rccl* r = rccl();
  if (!r) return fail(h, VOF_ESTATE, "RCCL (librccl.so.1) could not be loaded");
  const int W = VOF_HALO_ROWS(h->d.jacobi_iters);
  const bool lo = !h->g.wall_lo, hi = !h->g.wall_hi;  // interior edges
  const bool loop = (flags & VOF_COMM_LOOPBACK) != 0;
  if (lo && h->d.own_lo - W < h->d.row_lo) return fail(h, VOF_EINVAL, "fewer than VOF_HALO_ROWS rows stored below own_lo");
  if (hi && h->d.own_hi + W > h->d.row_hi) return fail(h, VOF_EINVAL, "fewer than VOF_HALO_ROWS rows stored above own_hi");
  if (h->d.own_hi - h->d.own_lo + 1 < W) return fail(h, VOF_EINVAL, "strip thinner than VOF_HALO_ROWS");
  if (!loop && ((lo && rank == 0) || (hi && rank == world - 1) || (!lo && rank != 0) || (!hi && rank != world - 1)))
    return fail(h, VOF_EINVAL, "rank does not match the strip: rank r of n owns the r-th row range from the left wall");
  HIPCHK(h, hipSetDevice(h->device));
  RcclId uid;
  memcpy(&uid, id, sizeof(uid));
  NCCLCHK(h, r->CommInitRank(&h->comm, world, uid, rank));
  HIPCHK(h, hipStreamCreateWithFlags(&h->cstream, hipStreamNonBlocking));
  HIPCHK(h, hipEventCreateWithFlags(&h->ev_ready, hipEventDisableTiming));
  HIPCHK(h, hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming));
  for (int k = 0; k < 3; ++k) HIPCHK(h, hipEventCreateWithFlags(&h->ev_fork[k], hipEventDisableTiming));
  // Capturing the send/recv groups into the step graph is verified with RCCL 2.27.7 (ROCm 7.2);
  // 2.26.6 (the copy bundled with PyTorch 2.10 + ROCm 7.0) crashes in hipStreamEndCapture.
  h->xchg_graph = r->version >= 22707 ? 1 : 0;
  const char* ev = getenv("VOF2D_XCHG_GRAPH");
  if (ev) h->xchg_graph = atoi(ev);
  h->xchg_steps = 0;
  h->comm_rank = rank;
  h->comm_world = world;
  // loopback (self-test on one GPU): both neighbours are this rank; RCCL pairs the k-th send to a
  // peer with the k-th receive from it, so each halo receives the W owned rows next to it
  h->peer_lo = lo ? (loop ? rank : rank - 1) : -1;
  h->peer_hi = hi ? (loop ? rank : rank + 1) : -1;
  return VOF_OK;
}
int vof_comm_allreduce_max(vof2d_handle h, double* value) {
  if (!h || !value) return VOF_EINVAL;
  if (!h->comm) return fail(h, VOF_ESTATE, "vof_comm_init has not been called");
  Rccl* r = rccl();
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->d_red) HIPCHK(h, hipMalloc(&h->d_red, sizeof(double)));
  // on the compute stream: ordered after everything enqueued so far, so it doubles as a barrier
  HIPCHK(h, hipMemcpyAsync(h->d_red, value, sizeof(double), hipMemcpyHostToDevice, h->stream));
  NCCLCHK(h, r->AllReduce(h->d_red, h->d_red, 1, /*ncclFloat64*/ 8, /*ncclMax*/ 2, h->comm, h->stream));
  HIPCHK(h, hipMemcpyAsync(value, h->d_red, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return VOF_OK;
}
int vof_comm_info(vof2d_handle h, int32_t* rccl_version, int32_t* graph_capture) {
  if (!h) return VOF_EINVAL;
  Rccl* r = rccl();
  if (rccl_version) *rccl_version = r ? r->version : 0;
  if (graph_capture) *graph_capture = (h->comm && h->xchg_graph && !(h->d.flags & VOF_FLAG_NO_GRAPH)) ? 1 : 0;
  return VOF_OK;
}
int vof_comm_destroy(vof2d_handle h) {
  if (!h) return VOF_EINVAL;
  comm_teardown(h);
  return VOF_OK;
}
static unsigned field_mask_ok(uint32_t mask) { return mask != 0 && (mask & ~15u) == 0; }
int vof_comm_exchange(vof2d_handle h, uint32_t field_mask) {
  if (!h) return VOF_EINVAL;
  if (!h->comm) return fail(h, VOF_ESTATE, "vof_comm_init has not been called");
  if (!field_mask_ok(field_mask)) return fail(h, VOF_EINVAL, "field_mask: VOF_XCHG_F | _U | _V | _P");
  HIPCHK(h, hipSetDevice(h->device));
  settle_ghosts(h);
  int rc = comm_post(h, field_mask);
  return rc ? rc : comm_join(h);
}
}  // extern "C"
namespace {
// One step with its exchanges on (compute stream, communication stream).  mode 0: one exchange of
// all four fields after the step; 1: each field leaves as soon as it is final (p after phase 0,
// u, v after phase 1, F after phase 2); 2: like 1, and F's edge bands are produced first so that F
// travels under the rest of the second sweep; 3: p, u, v together after phase 1, F after phase 2
// (one fork less).  Enqueued eagerly or under stream capture.
template <typename T>
int enqueue_step_exchange(vof2d_ctx* h, int mode) {
  int rc;
  // lean phases (no boundary launch inside): the rows travel with whatever ghost columns they
  // have, and one set_bc<u,v,F,p> over all stored rows -- owned and received alike -- follows the
  // join.  Only reached on steps that start with consistent F ghosts (vof_step_exchange).
  // With virtual ghosts (see enqueue_step) even that launch goes: the rows travel with stale ghost
  // columns and the next step's k_momentum forms the ones it reads, for owned and received rows alike.
  const bool lean = h->fuse_correct && h->fuse_momentum;
  const bool virt = lean && h->virtual_ghosts;
  enqueue_phase<T>(h, 0, h->istep, false, lean, virt, (int)(h->istep & 1));
  if (mode == 4) {
    // fused transport (update_uv + both sweeps in one pass), edge bands first: p, u, v and F (from
    // the twin buffer) leave as soon as the bands exist and travel under the transport of the
    // remaining rows
    const bool y_first = (h->istep % 2 == 0);
    transport_part<T>(h, y_first, kEdgeBands);
    // one group for all four fields: p has been final since the pressure solve, but a separate
    // fork for it costs more (a 6-12 us gap on the compute queue) than its 1/4 of the bytes
    if ((rc = comm_post(h, VOF_XCHG_P | VOF_XCHG_F | VOF_XCHG_U | VOF_XCHG_V, /*f_in_twin=*/true, 1))) return rc;
    transport_part<T>(h, y_first, kRest);
    swap_F(h);
    if ((rc = comm_join(h))) return rc;
    if (!virt) L<T>::template set_bc<BC_ALL>(h);
    return VOF_OK;
  }
  if ((mode == 1 || mode == 2) && (rc = comm_post(h, VOF_XCHG_P, false, 0))) return rc;   // p is final
  enqueue_phase<T>(h, 1, h->istep, false, lean);
  if (mode && (rc = comm_post(h, mode == 3 ? (VOF_XCHG_P | VOF_XCHG_U | VOF_XCHG_V) : (VOF_XCHG_U | VOF_XCHG_V), false, 1))) return rc;  // u, v are final
  if (mode == 2) {
    enqueue_phase<T>(h, VOF_PHASE_TRANSPORT_EDGES, h->istep, false, lean);
    if ((rc = comm_post(h, VOF_XCHG_F, /*f_in_twin=*/true, 2))) return rc;
    enqueue_phase<T>(h, VOF_PHASE_TRANSPORT_REST, h->istep, false, lean);
  } else {
    enqueue_phase<T>(h, 2, h->istep, false, lean);
    if ((rc = comm_post(h, mode ? VOF_XCHG_F : (VOF_XCHG_F | VOF_XCHG_U | VOF_XCHG_V | VOF_XCHG_P), false, 2))) return rc;
  }
  if ((rc = comm_join(h))) return rc;           // halos complete before the next step
  if (lean && !virt) L<T>::template set_bc<BC_ALL>(h);
  return VOF_OK;
}
}  // namespace
extern "C" {
int vof_step_exchange(vof2d_handle h, int64_t nsteps, int32_t overlap) {
  if (!h || nsteps < 0 || overlap < 0 || overlap > 4) return VOF_EINVAL;
  if (!h->comm) return fail(h, VOF_ESTATE, "vof_comm_init has not been called");
  if (h->next_phase != 0) return fail(h, VOF_ESTATE, "a phased step (vof_step_phase) is in progress");
  HIPCHK(h, hipSetDevice(h->device));
  const bool want_graph = !(h->d.flags & VOF_FLAG_NO_GRAPH);
  for (int64_t s = 0; s < nsteps; ++s) {
    // the captured step leaves the ghost cells virtual (if the handle does that at all); every other
    // way through this loop wants them settled first
    const bool captured_path = want_graph && h->xchg_graph && h->xchg_steps > 0 && !h->f_ghosts_dirty && !h->uv_ghosts_dirty;
    const bool virt = captured_path && h->fuse_correct && h->fuse_momentum && h->virtual_ghosts;
    if (!virt) settle_ghosts(h);
    h->istep += 1;
    const int par = (int)(h->istep & 1);
    int rc;
    // The first step of a communicator runs eagerly: RCCL sets its peer connections up on first
    // use, which must not happen inside a capture.  After that the whole step -- kernels on the
    // compute stream, the send/recv groups forked onto the communication stream, the join -- is
    // one hipGraph per (sweep order, mode): one launch per step instead of four graph launches
    // and three RCCL group launches (~100 us of host time each).
    const int ori = h->fld[fF] == h->f_home ? 0 : 1;
    const bool one_swap = overlap == 4;   // the fused transport swaps the F / twin pair once per step
    if (captured_path) {
      if (!h->gxchg[par][overlap][ori]) {
        void* keep[NFIELDS];
        memcpy(keep, h->fld, sizeof(keep));
        hipGraph_t graph = nullptr;
        const bool dbg = getenv("VOF2D_DEBUG") != nullptr;
        if (dbg) fprintf(stderr, "[vof2d] capturing step + exchange (parity %d, mode %d)\n", par, overlap);
        hipError_t e = hipStreamBeginCapture(h->stream, hipStreamCaptureModeRelaxed);
        rc = VOF_OK;
        if (e == hipSuccess) {
          DISPATCH_T(h, rc = enqueue_step_exchange<double>(h, overlap), rc = enqueue_step_exchange<float>(h, overlap));
          if (dbg) fprintf(stderr, "[vof2d]   enqueued (rc %d), ending capture\n", rc);
          e = hipStreamEndCapture(h->stream, &graph);
          if (dbg) fprintf(stderr, "[vof2d]   capture ended: %s\n", hipGetErrorString(e));
        }
        if (e == hipSuccess && rc == VOF_OK && graph) e = hipGraphInstantiate(&h->gxchg[par][overlap][ori], graph, nullptr, nullptr, 0);
        if (graph) (void)hipGraphDestroy(graph);
        if (one_swap) memcpy(h->fld, keep, sizeof(keep));   // capturing swapped the host's view; the replay below redoes it
        if (e != hipSuccess || rc != VOF_OK || !h->gxchg[par][overlap][ori]) {
          // this RCCL / runtime cannot capture the exchange: keep going with eager launches
          (void)hipGetLastError();
          memcpy(h->fld, keep, sizeof(keep));
          h->gxchg[par][overlap][ori] = nullptr;
          h->xchg_graph = 0;
          // a capture that failed after the fork may leave the communication stream inside the
          // invalidated capture: the eager launches below need a working one
          hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
          if (hipStreamIsCapturing(h->cstream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            (void)hipStreamDestroy(h->cstream);
            h->cstream = nullptr;
            if (hipStreamCreateWithFlags(&h->cstream, hipStreamNonBlocking) != hipSuccess)
              return fail(h, VOF_EHIP, "cannot recreate the communication stream after a failed capture");
          }
          if (getenv("VOF2D_DEBUG")) fprintf(stderr, "[vof2d] exchange graph capture failed (%s / %s): eager\n", hipGetErrorString(e), h->err);
        }
      }
      if (h->gxchg[par][overlap][ori]) {
        HIPCHK(h, hipGraphLaunch(h->gxchg[par][overlap][ori], h->stream));
        if (one_swap) swap_F(h);
        h->xchg_steps += 1;
        h->xchg_graph_steps += 1;
        h->ghosts_virtual = virt;
        continue;
      }
    }
    h->istep -= 1;  // vof_step_phase(0) advances it
    const int eo = overlap == 4 ? 1 : overlap;   // eager steps (the first of a communicator, ...) of mode 4 run as mode 1
    if ((rc = vof_step_phase(h, 0))) return rc;
    if ((eo == 1 || eo == 2) && (rc = comm_post(h, VOF_XCHG_P))) return rc;
    if ((rc = vof_step_phase(h, 1))) return rc;
    if (eo && (rc = comm_post(h, eo == 3 ? (VOF_XCHG_P | VOF_XCHG_U | VOF_XCHG_V) : (VOF_XCHG_U | VOF_XCHG_V)))) return rc;
    if (eo == 2) {
      if ((rc = vof_step_phase(h, VOF_PHASE_TRANSPORT_EDGES))) return rc;
      if ((rc = comm_post(h, VOF_XCHG_F, /*f_in_twin=*/true))) return rc;
      if ((rc = vof_step_phase(h, VOF_PHASE_TRANSPORT_REST))) return rc;
    } else {
      if ((rc = vof_step_phase(h, 2))) return rc;
      if ((rc = comm_post(h, eo ? VOF_XCHG_F : (VOF_XCHG_F | VOF_XCHG_U | VOF_XCHG_V | VOF_XCHG_P)))) return rc;
    }
    if ((rc = comm_join(h))) return rc;
    h->xchg_steps += 1;
  }
  return VOF_OK;
}

// ---- self-test of the exact constant-denominator division (vof2d_kernels.h div_by_const) against
// the hardware IEEE division, on adversarial numerators: subnormal quotients at and next to the
// midpoints of the subnormal grid (the double-rounding case), tiny / huge / special values.
}  // extern "C"
namespace {
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
extern "C" {
int vof_selftest_division(int32_t dtype, int64_t n, uint64_t seed, void* a_out, void* b_out, void* q_out) {
  if (!a_out || !b_out || !q_out || n < 1 || (dtype != VOF_F64 && dtype != VOF_F32)) return VOF_EINVAL;
  const size_t bytes = (size_t)n * (dtype == VOF_F64 ? 8 : 4);
  char* dev = nullptr;
  if (hipMalloc(&dev, 3 * bytes) != hipSuccess) { (void)hipGetLastError(); return VOF_ENOMEM; }
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if (dtype == VOF_F64)
    hipLaunchKernelGGL(k_selftest_division<double>, dim3(blocks), dim3(256), 0, 0, seed, n, (double*)dev,
                       (double*)(dev + bytes), (double*)(dev + 2 * bytes));
  else
    hipLaunchKernelGGL(k_selftest_division<float>, dim3(blocks), dim3(256), 0, 0, seed, n, (float*)dev,
                       (float*)(dev + bytes), (float*)(dev + 2 * bytes));
  hipError_t e = hipMemcpy(a_out, dev, bytes, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(b_out, dev + bytes, bytes, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(q_out, dev + 2 * bytes, bytes, hipMemcpyDeviceToHost);
  (void)hipFree(dev);
  return e == hipSuccess ? VOF_OK : VOF_EHIP;
}
const char* vof_last_error(vof2d_handle h) { return h ? h->err : "null handle"; }
const char* vof_backend(void) { return "hip-gfx950"; }

}  // extern "C"
