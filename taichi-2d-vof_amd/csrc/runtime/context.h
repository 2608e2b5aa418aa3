// runtime/context.h -- field ids, constants folded like the reference does, the handle (vof2d_ctx), error helpers
//
// Part of the host-side runtime of libvof2d_hip.so; included (once, in this order) by vof2d_api.hip:
// context.h, launches.h, schedule.h, comm.h, selftest.h.  Everything here has internal linkage.
#pragma once
#include <vector>

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

#include "../../../include/vof2d.h"
#include "../vof2d_kernels.h"

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
  c.dt_rho_l = c.dt / c.rho_l; c.dt_rho_g = c.dt / c.rho_g;   // IEEE quotients in T, as the device's dt / r
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
  int tb_general = 0;   // force the general (dx != dy) fused Jacobi kernel (tests: both forms on square cells)
  int tb_adapt = 1;     // fused steps: shorter chunks on the tile columns the tiny-value front is crossing (k_jacobi_tb)
  unsigned long long* d_tbmask = nullptr;  // work plan of k_jacobi_tb (TbPlan): 2 x TB_BANDS mask words, then the plan (1 + waves entries)
  long tbplan_cap = 0;                     // waves the plan area holds
  int fctx_rows = 0;    // rows per wave chunk of k_fct_x (0 = heuristic, at most 16)
  int fctx_corr_rows = 0;  // ... of its update_uv-carrying form (0 = same rule)
  int fuse_transport = 1;  // vof_step on a full domain: update_uv and both FCT sweeps in one kernel (k_transport)
  int band_rows = 4;       // rows per wave chunk of the edge-band launch of the fused transport (strips)
  int buf_stores = 7;      // range-checked buffer stores where the grid allows (buffer_stores_ok): bit 0 k_momentum, bit 1 k_jacobi_tb, bit 2 k_tm's momentum wave
  int virtual_ghosts = 1;  // ... without the step's set_BC launch (k_momentum forms the ghost cells it reads)
  void* f_home = nullptr;  // the buffer fld[fF] pointed to at creation (orientation of the F / twin pair)
  void* us_home = nullptr; // ... fld[fUS] (the u*, v* pair alternates with mx, my in the k_tm forms) and fld[fP] (k_jacobi_pair alternates p / pt)
  void* p_home = nullptr;
  int phase_graph_ori = 0; // orientation the gphase / gxchg graphs were captured in
  hipGraphExec_t gexec[2][2] = {};  // whole step, [istep parity][F in its home buffer ? 0 : 1]
  // several consecutive steady-state steps of a full domain as ONE graph (step_batch[b] steps, an even number: the
  // F / twin pair and the step parity are back where they started): a graph launch leaves ~9 us of idle queue
  // behind it, which one launch per step pays every step (vof_step)
  static constexpr int kStepBatches = 3;
  static constexpr int kTuneBatch = 1;     // the batch size the two forms are timed with (fuse_tm = -2)
  // (the k_tm form has one plain k_momentum and one plain k_transport per batch: 4096^2 0.5256 ms/step in batches of 8, 0.5193 of 16,
  // 0.5178 of 32, 0.5346 of 4; knob "batch_steps" sets the first)
  int step_batch[kStepBatches] = {16, 8, 2};
  hipGraphExec_t gbatch[kStepBatches][2][2] = {};   // [batch size][parity of the first step][orientation]
  // knob "overlap_halves": the batch graphs run every kernel of a step as two launches, on the rows above and below a
  // boundary that moves up by kHalvesDrift rows from kernel to kernel, the upper chain on `stream`, the lower on
  // `chain_streams`; a lower launch waits for the upper launch of the kernel before it only (enqueue_steps_halves)
  // knob "fuse_tm": the batch graphs run k_transport + the next step's k_momentum as one kernel (k_tm, kernels/fused_tm.h):
  // 0 never, 1 wherever the schedule allows, -1 (default) on large fp64 grids by a rule on the state (the share of gas
  // cells: decide_batch_form_by_rule), -2 (exploration) after timing both forms on the handle's own data: four 8-step
  // batches alternate between the forms, the faster one stays (vof_step)
  int fuse_tm = -1;
  double gas_share = -1.0;   // what the rule saw (get_param "gas_share")
  unsigned long long* h_gas = nullptr;   // pinned host word the count of exact-zero cells of F lands in (post_gas_count)
  hipEvent_t ev_gas = nullptr;           // ... recorded behind the copy
  bool gas_pending = false;              // a count of the current F is in flight or has landed, not yet read
  bool tm_broken = false;    // the k_tm batch graphs could not be captured: the other form stays
  bool alt_dirty = false;    // a verb or a field write left something in mx / my (the second u*, v* pair of the k_tm forms): cleared at the head of a strip call (tm5_head)
  bool ahead = false;        // u*, v*, rhs hold the predictor of step istep + 1 (the chained k_tm batches; settle_ahead)
  bool tm_rhs_alt = false;   // the k_tm launches being enqueued write rhs into the kappa array (the caller swaps the views)
  int64_t tm_chained = 0;    // k_tm batches that started without a k_momentum launch (counter "tm_chained_batches")
  int jpair = 1;             // knob "jacobi_pair": the k_tm batch graphs run each two five-sweep launches as one k_jacobi_pair launch
  int jpair_rows = 0;        // rows per pair chunk (0 = one residency round of pairs)
  int solve_pairs = -1;      // knob "solve_pairs": jacobi_n (the residual-terminated solve, the verbs, the last step of a strip call) runs ten sweeps per launch as k_jacobi_pair where it applies (-1: from 4 M cells on, solve_pairs_on)
  // knobs "pair_slow10", "tb_slow10": what the planner takes a row of a reported band to cost, in tenths of an ordinary row.
  // k_jacobi_pair, 4096^2 dam-break, us per launch in steps 301-350 / ms per step over steps 61-660: 20 146.6 / 0.4454, 26 132.0 / 0.4355,
  // 32 127.2 / 0.4312, 40 128.4 / 0.4328, 50 129.8 / 0.4338 (the cold tier costs more against the FAST sub-iterations than it did
  // against round 4's rows)
  int pair_slow10 = 32;
  int tb_slow10 = 20;
  int pair_vec4 = 0;         // knob "pair_vec4": fp32 pair kernels with four columns per lane (256-column tiles)
  bool jpair_active = false; // the launches being enqueued are k_jacobi_pair's (tb_plan describes their geometry)
  bool jpair_captured = false;   // the k_tm batch graphs the handle holds contain k_jacobi_pair launches
  int64_t pair_launches = 0; // k_jacobi_pair launches replayed (counter "pair_launches")
  hipGraphExec_t gbatch_tm[kStepBatches][2][2] = {};   // the k_tm form of gbatch
  int tune_n = 0;            // timed batches so far (even: chains / plain, odd: k_tm); 4: ready to decide; 5: decided
  int tm_choice = 0;         // the form that stays -- until the next timing: every tune_period batches the two forms are timed
  int tune_period = 2048;    // again (a flow changes character over 16 000 steps; four alternating batches cost next to nothing)
  int tune_age = 0;          // batches since the last decision
  bool tm_decided = false;   // a decision has been made since the graphs were built
  hipEvent_t tune_ev[8] = {};
  float tune_ms[2] = {0.f, 0.f};
  int64_t tm_steps = 0;      // steps replayed from k_tm batch graphs (counter "tm_steps")
  int tm_rows = 0;          // rows per pair chunk of k_tm (0 = 32)
  int halves = -1;   // -1: where it pays (halves_eligible), 0: never, 1: wherever the schedule allows
  std::vector<hipStream_t> chain_streams;   // streams of the chains below the first
  bool halves_captured[kStepBatches] = {};   // the batch graphs of size step_batch[b] the handle holds were captured in this form
  int64_t halves_steps = 0;       // steps replayed from them (counter "halves_steps")
  std::vector<hipEvent_t> hev;
  bool batching = true;         // false after a failed capture of a batch: one graph launch per step from then on (build_step_batches)
  hipGraphExec_t gphase[5] = {};  // phase 0, then phases 1, 2 x istep parity (slot 2 * phase - 1 + parity)
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
  hipGraphExec_t gxchg2[2][2] = {};     // TWO mode-4 steps + exchanges per launch, [parity of the first][orientation] (vof_step_exchange)
  hipGraphExec_t gxchg5[16] = {};       // TWO middle steps of mode 5 per launch, [parity | F orientation << 1 | u*, v* orientation << 2 | p orientation << 3]
  int xchg5_graph = 1;                  // ... while this RCCL / runtime captures them
  int xchg_pair = 1;                 // 0 after a failed capture of a pair: one step per launch
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

}  // namespace
