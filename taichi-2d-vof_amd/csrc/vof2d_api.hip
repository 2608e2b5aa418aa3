// vof2d_api.hip -- C ABI (include/vof2d.h) over the gfx950 kernels.
//
// Host-side runtime of the drop-in: owns the device arena, the HIP stream, the
// per-step launch schedule (eager or hipGraph replay) and the pitched
// host<->device copies behind to_numpy()/from_numpy().  No CPU compute path
// exists here: every verb is a kernel launch.
//
// This file holds the extern "C" entry points only.  The runtime behind them:
//   runtime/context.h    the handle, constants, chunk-length heuristics
//   runtime/launches.h   one launch wrapper per kernel
//   runtime/schedule.h   the per-step schedule, ghost-cell bookkeeping, graphs
//   runtime/comm.h       strips over RCCL (bound with dlopen)
//   runtime/selftest.h   device side of the division self-test
#include "runtime/context.h"
#include "runtime/launches.h"
#include "runtime/schedule.h"
#include "runtime/comm.h"
#include "runtime/selftest.h"
#ifdef VOF_WAVE_TIMES
#include "runtime/diag.h"
#endif

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
  h->nty = (d->ny + (W - 2 * TileHalo::transport) - 1) / (W - 2 * TileHalo::transport);
  const int align = 128 / (int)h->esz;  // elements per 128 bytes
  g.col0 = align - 1;                   // j = 1 lands on a 128-byte boundary
  // furthest column any lane touches: the overlapped tiles of k_fct_y / k_jacobi_tb start at most
  // H <= 12 columns left of j = 1 and their last tile may run a full tile past ny.
#ifdef VOF_PAIR_VEC4   // (experiment: the fp32 pair kernels with four columns per lane, 256-column tiles)
  long maxcol = (long)d->ny + (d->dtype == VOF_F32 ? 256 : W) + 16;
#else
  long maxcol = (long)d->ny + W + 16;
#endif
  g.pitch = ((g.col0 + maxcol + 1 + align - 1) / align) * align;
  const size_t nrows = (size_t)(d->row_hi - d->row_lo + 1);
  h->field_elems = nrows * (size_t)g.pitch + (size_t)align;  // + one 128-byte tail pad
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
    if (const char* ev = getenv("VOF2D_OVERLAP_HALVES")) h->halves = atoi(ev);   // (profiling runs: per-kernel counters want one kernel at a time)
    if (const char* ev = getenv("VOF2D_FUSE_TM")) h->fuse_tm = atoi(ev);
#ifdef VOF_ARENA_EXP   // placement experiment (tools/probes/arena_modes.py): shift of the whole arena, extra bytes between fields
    const size_t shift_ = getenv("VOF2D_ARENA_SHIFT") ? (size_t)atoll(getenv("VOF2D_ARENA_SHIFT")) : 0;
    const size_t skew_ = getenv("VOF2D_FIELD_SKEW") ? (size_t)atoll(getenv("VOF2D_FIELD_SKEW")) : 0;
    const size_t bytes = (h->field_elems * h->esz + skew_) * NFIELDS + shift_;
    if (hipMalloc(reinterpret_cast<void**>(&h->arena), bytes) != hipSuccess) { rc = VOF_ENOMEM; break; }
    if (hipMemsetAsync(h->arena, 0, bytes, h->stream) != hipSuccess) { rc = VOF_EHIP; break; }
    for (int k = 0; k < NFIELDS; ++k) h->fld[k] = h->arena + shift_ + (size_t)k * (h->field_elems * h->esz + skew_);
#else
    const size_t bytes = h->field_elems * h->esz * NFIELDS;
    if (hipMalloc(reinterpret_cast<void**>(&h->arena), bytes) != hipSuccess) { rc = VOF_ENOMEM; break; }
    if (hipMemsetAsync(h->arena, 0, bytes, h->stream) != hipSuccess) { rc = VOF_EHIP; break; }
    for (int k = 0; k < NFIELDS; ++k) h->fld[k] = h->arena + (size_t)k * h->field_elems * h->esz;
#endif
    h->f_home = h->fld[fF];
    h->us_home = h->fld[fUS];
    h->p_home = h->fld[fP];
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
  for (hipEvent_t e : h->hev) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->tune_ev) if (e) (void)hipEventDestroy(e);
  for (hipStream_t st : h->chain_streams) (void)hipStreamDestroy(st);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->ev_gas) (void)hipEventDestroy(h->ev_gas);
  if (h->h_gas) (void)hipHostFree(h->h_gas);
  comm_teardown(h);
  if (h->vis) (void)hipFree(h->vis);
#ifdef VOF_SHORTCUT_STATS
  {
    unsigned long long a[16] = {};
    if (hipMemcpyFromSymbol(a, HIP_SYMBOL(vof::vof_stats), sizeof(a)) == hipSuccess && a[0] + a[3])
      fprintf(stderr, "[vof2d] shortcut stats (wave-rows): momentum rows %llu flat-normals %llu no-force %llu | transport rows %llu gas %llu liquid %llu "
              "uniform-update_uv %llu x-pipe-zero-bypass %llu x-stageB-skipped %llu x-stageD-clamp %llu | y rows swept %llu zero-flux %llu\n",
              a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11]);
  }
#endif
  if (h->d_courant) (void)hipFree(h->d_courant);
  if (h->d_tbmask) (void)hipFree(h->d_tbmask);
  if (h->arena) (void)hipFree(h->arena);
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return VOF_OK;
}

static bool post_gas_count(vof2d_ctx* h);   // (the batch-form rule of vof_step, below)
int vof_set_init_F(vof2d_handle h, int32_t ic) {
  if (!h) return VOF_EINVAL;
  if (ic < 1 || ic > 3) return fail(h, VOF_EINVAL, "ic must be 1, 2 or 3 (2dvof.py:13)");
  settle_ghosts(h);
  DISPATCH_T(h, L<double>::init_F(h, ic), L<float>::init_F(h, ic));
  h->f_ghosts_dirty = true;
  if (h->fuse_tm == -1) {   // (the batch-form rule looks at the new F: the count is taken now, asynchronously, and read by the first batched step)
    h->tm_decided = false;
    h->gas_pending = false;
    if (tm_by_rule(h)) (void)post_gas_count(h);
  }
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
  h->alt_dirty = true;
  return ensure_ok(h);
}
int vof_advect_upwind(vof2d_handle h) {
  if (!h) return VOF_EINVAL;
  settle_ghosts(h);
  DISPATCH_T(h, L<double>::predictor<true>(h), L<float>::predictor<true>(h));
  h->alt_dirty = true;
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
  for (int k = 0; k < 4 * vof2d_ctx::kStepBatches; ++k) any = any || h->gbatch[k / 4][(k / 2) % 2][k % 2] || h->gbatch_tm[k / 4][(k / 2) % 2][k % 2];
  for (int k = 0; k < 5; ++k) any = any || h->gphase[k];
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

// The batch graphs (kStepBatch[b] steady-state steps per launch) of the (parity, orientation) pair the current step
// finds and of the pair the next step will find -- the two pairs a run alternates between; a handle whose parity
// was moved alone (vof_set_istep) gets the other two on its next steady-state step.  Captures enqueue nothing.  Any
// failure on the way ends the capture, puts the F / twin pair back, switches batching off for the handle and leaves
// the single-step graphs (or eager launches) to carry on: never an error of vof_step.
// steps per graph launch of batch size b: the chained k_tm batches pay one u, v store per batch and nothing else, so
// their largest is twice the other form's (whose chains drift kHalvesDrift rows per launch: halves_prepare)
static int batch_steps(const vof2d_ctx* h, int variant, int b) {
  return (variant && b == 0) ? 2 * h->step_batch[0] : h->step_batch[b];
}
static void build_step_batches(vof2d_ctx* h, int variant /* 0: chains or the plain sequence, 1: k_tm */) {
  auto& GB = variant ? h->gbatch_tm : h->gbatch;
  void* const f0 = h->fld[fF];
  void* const f1 = h->fld[fF2];
  bool ok = true;
  for (int c = 0; c < 2 && ok; ++c) {
    if (c) swap_F(h);                                   // the pair as the NEXT step will find it
    const int64_t first = h->istep + c;
    const int ori_c = h->fld[fF] == h->f_home ? 0 : 1;
    for (int b = 0; b < vof2d_ctx::kStepBatches && ok; ++b) {
      hipGraphExec_t& slot = GB[b][(int)(first & 1)][ori_c];
      if (slot) continue;
      hipGraph_t graph = nullptr;
      const bool fused_tm = variant == 1;
      const bool chains = !fused_tm && halves_eligible(h, h->step_batch[b]) && halves_prepare(h, h->step_batch[b]);
      if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { ok = false; break; }
      bool enq = true;
      if (chains) {
        h->halves_captured[b] = true;
        DISPATCH_T(h, enq = enqueue_steps_halves<double>(h, first, h->step_batch[b]), enq = enqueue_steps_halves<float>(h, first, h->step_batch[b]));
      } else if (fused_tm) {
        DISPATCH_T(h, enqueue_steps_tm<double>(h, first, batch_steps(h, 1, b)), enqueue_steps_tm<float>(h, first, batch_steps(h, 1, b)));
      } else
        for (int k = 0; k < h->step_batch[b]; ++k)
          DISPATCH_T(h, enqueue_step<double>(h, first + k, true, true), enqueue_step<float>(h, first + k, true, true));
      hipError_t e = hipStreamEndCapture(h->stream, &graph);          // (always: the stream must leave capture mode)
      if (e == hipSuccess && !enq) e = hipErrorUnknown;
      if (e == hipSuccess) e = hipGraphInstantiate(&slot, graph, nullptr, nullptr, 0);
      if (graph) (void)hipGraphDestroy(graph);
      if (e != hipSuccess) { slot = nullptr; ok = false; break; }
      (void)hipGraphUpload(slot, h->stream);     // so that the first replay -- possibly inside a timed region -- does not pay for it
    }
    h->fld[fF] = f0;                             // whatever the captured steps did to the host's view of the pair
    h->fld[fF2] = f1;
  }
  if (!ok) {
    (void)hipGetLastError();
    for (int b = 0; b < vof2d_ctx::kStepBatches; ++b)
      for (int k = 0; k < 4; ++k)
        if (GB[b][k >> 1][k & 1]) { (void)hipGraphExecDestroy(GB[b][k >> 1][k & 1]); GB[b][k >> 1][k & 1] = nullptr; }
    // the k_tm form failing leaves the other form's batch graphs in use; only when those fail is it one graph launch per step
    if (variant) h->tm_broken = true; else h->batching = false;
    if (getenv("VOF2D_DEBUG")) fprintf(stderr, "[vof2d] step batches (%s) could not be captured: %s\n", variant ? "k_tm form" : "chains / plain",
                                       variant ? "the other form stays" : "one graph launch per step");
  }
}

// Which form of the batch graphs a large fp64 full domain runs (knob fuse_tm = -1, the default): a RULE on the state, so
// that two handles on the same data always run the same schedule.  k_tm + k_jacobi_pair win where most rows are cheap for
// the transport pipeline (gas: the x pipeline bypasses itself, the y stage is skipped) and lose where they are not -- 4096^2
// dam-break (5/6 gas) 0.49 against 0.56 ms/step for the chains, 4096^2 rising bubble (2 % gas) 0.81 against 0.61 -- so the
// rule is the share of exact-zero cells of F when the handle first batches steps (and again after F was replaced from
// outside): one small kernel and one 8-byte read-back, where the graphs are being captured anyway.
// The count is POSTED (kernel + 8-byte copy into pinned host memory + event, all asynchronous) where F is replaced as a
// whole -- vof_set_init_F -- or, failing that, by the first step that needs it; vof_step only waits for the event, which
// after set_init_F has long fired: no device sync inside a timed vof_step.  Anything that fails on the way (a caller's
// stream under capture, no pinned memory) leaves the handle undecided and on the other form: never an error of vof_step.
static bool post_gas_count(vof2d_ctx* h) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(h->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return false; }
  if (!h->h_gas && hipHostMalloc(reinterpret_cast<void**>(&h->h_gas), sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); h->h_gas = nullptr; return false; }
  if (!h->ev_gas && hipEventCreateWithFlags(&h->ev_gas, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); h->ev_gas = nullptr; return false; }
  unsigned long long* cnt = h->d_courant + 3;
  bool ok = hipMemsetAsync(cnt, 0, sizeof(*cnt), h->stream) == hipSuccess;
  const unsigned blocks = (unsigned)(h->g.ihi - h->g.ilo + 1 < 2048 ? h->g.ihi - h->g.ilo + 1 : 2048);
  if (h->d.dtype == VOF_F64) hipLaunchKernelGGL(k_gas_cells<double>, dim3(blocks), dim3(256), 0, h->stream, h->g, (const double*)F_<double>(h, fF), cnt);
  else hipLaunchKernelGGL(k_gas_cells<float>, dim3(blocks), dim3(256), 0, h->stream, h->g, (const float*)F_<float>(h, fF), cnt);
  ok = ok && hipMemcpyAsync(h->h_gas, cnt, sizeof(*cnt), hipMemcpyDeviceToHost, h->stream) == hipSuccess;
  ok = ok && hipEventRecord(h->ev_gas, h->stream) == hipSuccess;
  if (!ok) (void)hipGetLastError();
  h->gas_pending = ok;
  return ok;
}
static void decide_batch_form_by_rule(vof2d_ctx* h) {
  if (!h->gas_pending && !post_gas_count(h)) return;
  h->gas_pending = false;
  if (hipEventSynchronize(h->ev_gas) != hipSuccess) { (void)hipGetLastError(); return; }
  const unsigned long long n = *h->h_gas;
  h->gas_share = (double)n / ((double)(h->g.ihi - h->g.ilo + 1) * (double)h->g.ny);
  h->tm_choice = tm_choice_by_rule(h, h->gas_share);
  h->tm_decided = true;
  if (getenv("VOF2D_DEBUG")) fprintf(stderr, "[vof2d] batch form by rule: %.3f of the cells are gas -> %s\n", h->gas_share, h->tm_choice ? "k_tm" : "chains / plain");
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
    if (use_graph && lean && regular && virt) {
      // steady state of a full domain: as many of the remaining steps as possible in batches, one graph launch
      // each.  An even number of steps leaves the F / twin pair and the host's view of it where they were.
      // Parity and orientation flip together from step to step, so two (parity, orientation) pairs are
      // reachable; the batch graphs of both are captured the first time a steady-state step comes by (captures
      // enqueue nothing), so that no later call pays for an instantiation in the middle of a run.
      // Which form of the batch graphs: k_tm (variant 1) where the knob says so; with the knob on "auto" the handle
      // first times both on its own data -- four 8-step batches, alternating -- and keeps the faster (tm_auto).
      int variant = (h->fuse_tm > 0 && tm_eligible(h)) ? 1 : 0;
      bool timed = false;
      if (tm_by_rule(h)) {
        if (!h->tm_decided) decide_batch_form_by_rule(h);
        variant = h->tm_decided ? h->tm_choice : 0;   // (undecided -- the count could not be taken: the other form, and another try next call)
      } else if (tm_auto(h)) {   // fuse_tm = -2 (exploration): both forms timed on the handle's own data
        if (h->tune_n == 4) {
          bool done = hipEventSynchronize(h->tune_ev[7]) == hipSuccess;
          for (int k = 0; k < 4 && done; ++k) {
            float ms = 0.f;
            done = hipEventElapsedTime(&ms, h->tune_ev[2 * k], h->tune_ev[2 * k + 1]) == hipSuccess;
            h->tune_ms[k & 1] += ms;
          }
          h->tm_choice = (done && h->tune_ms[1] < 0.99f * h->tune_ms[0]) ? 1 : 0;
          h->tune_n = 5;
          h->tm_decided = true;
          if (!done) (void)hipGetLastError();
          if (getenv("VOF2D_DEBUG")) fprintf(stderr, "[vof2d] batch forms timed: %.3f ms (chains / plain) vs %.3f ms (k_tm) per 16 steps -> %s\n", h->tune_ms[0], h->tune_ms[1], h->tm_choice ? "k_tm" : "chains / plain");
        }
        if (h->tune_n == 5 && h->tune_period > 0 && h->tune_age >= h->tune_period) {   // time the forms again
          h->tune_n = 0; h->tune_age = 0; h->tune_ms[0] = h->tune_ms[1] = 0.f;
        }
        if (h->tune_n < 4) {
          timed = nsteps - s >= h->step_batch[vof2d_ctx::kTuneBatch];
          variant = timed ? (h->tune_n & 1) : (h->tm_decided ? h->tm_choice : 0);
        }
        else { variant = h->tm_choice; h->tune_age += 1; }
      }
      if (variant && h->tm_broken) variant = 0;
      if (h->batching && !(variant ? h->gbatch_tm : h->gbatch)[0][par][h->fld[fF] == h->f_home ? 0 : 1]) {
        build_step_batches(h, variant);
        if (variant && h->tm_broken) { variant = 0; timed = false; if (h->batching && !h->gbatch[0][par][h->fld[fF] == h->f_home ? 0 : 1]) build_step_batches(h, 0); }
      }
      auto& GB = variant ? h->gbatch_tm : h->gbatch;
      const int ori = h->fld[fF] == h->f_home ? 0 : 1;
      bool batched = false;
      for (int b = timed ? vof2d_ctx::kTuneBatch : 0; b < vof2d_ctx::kStepBatches && !batched; ++b) {   // (while the forms are being timed: batches of the timed size)
        const int K = batch_steps(h, variant, b);
        if (nsteps - s < K || !GB[b][par][ori]) continue;
        const bool time_it = timed && b == vof2d_ctx::kTuneBatch && h->batching;
        if (time_it) {
          for (int k = 0; k < 2; ++k)
            if (!h->tune_ev[2 * h->tune_n + k] && hipEventCreate(&h->tune_ev[2 * h->tune_n + k]) != hipSuccess) { h->istep -= 1; return fail(h, VOF_EHIP, "hipEventCreate"); }
          if (hipEventRecord(h->tune_ev[2 * h->tune_n], h->stream) != hipSuccess) { h->istep -= 1; return fail(h, VOF_EHIP, "hipEventRecord"); }
        }
        if (variant) {
          // the k_tm batches chain: each ends with the next step's predictor in place (enqueue_steps_tm); only the first
          // after anything else needs its k_momentum launched in front
          if (h->ahead) h->tm_chained += 1;
          else DISPATCH_T(h, enqueue_tm_head<double>(h, par), enqueue_tm_head<float>(h, par));
          h->ahead = true;
        } else h->ahead = false;
        if (hipGraphLaunch(GB[b][par][ori], h->stream) != hipSuccess) { h->istep -= 1; return fail(h, VOF_EHIP, "hipGraphLaunch of a step batch"); }
        if (time_it) {
          if (hipEventRecord(h->tune_ev[2 * h->tune_n + 1], h->stream) == hipSuccess) h->tune_n += 1;
          else (void)hipGetLastError();   // (the batch ran: this timing is lost, the steps are not)
        }
        h->istep += K - 1;
        s += K - 1;
        if (variant) { h->tm_steps += K; if (h->jpair_captured) h->pair_launches += (int64_t)K * (h->d.jacobi_iters / 10); }
        else if (h->halves_captured[b]) h->halves_steps += K;
        batched = true;
      }
      if (batched) {
        h->ghosts_virtual = true;
        continue;
      }
    }
    h->ahead = false;   // (this step forms its own predictor, into the host's view of u*, v*, rhs)
    if (use_graph && lean && regular) {
      // graphs bake the field pointers in: one per (parity, which buffer of the F / twin pair holds
      // F).  The two-kernel transport swaps the pair twice per step, the fused one once.
      const int ori = h->fld[fF] == h->f_home ? 0 : 1;
      const bool one_swap = h->g.wall_lo && h->g.wall_hi && h->fuse_transport;
      if (!h->gexec[par][ori]) {
        hipGraph_t graph = nullptr;
        HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        DISPATCH_T(h, enqueue_step<double>(h, h->istep, true, virt), enqueue_step<float>(h, h->istep, true, virt));
        HIPCHK(h, hipStreamEndCapture(h->stream, &graph));
        hipError_t e = hipGraphInstantiate(&h->gexec[par][ori], graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) {
          snprintf(h->err, sizeof(h->err), "hipGraphInstantiate: %s", hipGetErrorString(e));
          if (one_swap) swap_F(h);
          return VOF_EHIP;
        }
        (void)hipGraphUpload(h->gexec[par][ori], h->stream);
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
  for (int k = 0; k < 5; ++k) any = any || h->gphase[k];
  if (any) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int k = 0; k < 5; ++k)
      if (h->gphase[k]) { (void)hipGraphExecDestroy(h->gphase[k]); h->gphase[k] = nullptr; }
  }
  h->phase_graph_ori = ori;
  return VOF_OK;
}
int vof_step_phase(vof2d_handle h, int32_t phase) {
  if (!h) return VOF_EINVAL;
  if (phase < 0 || phase > 2) return fail(h, VOF_EINVAL, "phase must be 0, 1 or 2");
  if (phase != h->next_phase) return fail(h, VOF_ESTATE, "vof_step_phase must be called in the order 0, 1, 2");
  if (phase == 0) {
    settle_ghosts(h);
    int rc = match_phase_graph_orientation(h);
    if (rc) return rc;
    h->istep += 1;
  }
  h->next_phase = phase == 2 ? 0 : phase + 1;
  if (phase == 2) h->f_ghosts_dirty = h->uv_ghosts_dirty = false;   // the phases carry every set_BC of the step
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
  if (!captured_now && (phase == 1 || phase == 2)) swap_F(h);
  return VOF_OK;
}
int vof_get_istep(vof2d_handle h, int64_t* istep) {
  if (!h || !istep) return VOF_EINVAL;
  *istep = h->istep;
  return VOF_OK;
}
int vof_set_istep(vof2d_handle h, int64_t istep) {
  if (!h) return VOF_EINVAL;
  (void)settle_ahead(h);   // (the plan the last k_tm left is of the parity that was to follow)
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
  if (id == fF || id == fF2) { h->f_ghosts_dirty = true; if (h->fuse_tm == -1) { h->tm_decided = false; h->gas_pending = false; } }
  if (id == fMX || id == fMY) h->alt_dirty = true;
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
  // order, both ways: the copy runs on dst's stream after src's pending work, and whatever src enqueues next runs after
  // the copy -- a strip's next launches overwrite rows in place (p: the second five-sweep launch or the copy-back of an
  // odd launch count; rhs: k_tm) that a neighbour's pending copy may still have to read.  (Until round 6 only the first
  // half held: the differential fuzz of tests/test_fuzz_gpu.py met the other one in 2 of 1700 emulated strip runs,
  // both with five sweeps per step -- the shortest way from a copy to the next in-place write of p.)
  HIPCHK(dst, hipEventRecord(src->ev1, src->stream));
  HIPCHK(dst, hipStreamWaitEvent(dst->stream, src->ev1, 0));
  HIPCHK(dst, hipMemcpyAsync(reinterpret_cast<char*>(dst->fld[id]) + off_d,
                             reinterpret_cast<char*>(src->fld[id]) + off_s, bytes, hipMemcpyDeviceToDevice,
                             dst->stream));
  if (id == fF)
    HIPCHK(dst, hipMemcpyAsync(reinterpret_cast<char*>(dst->fld[fF2]) + off_d,
                               reinterpret_cast<char*>(src->fld[fF]) + off_s, bytes, hipMemcpyDeviceToDevice,
                               dst->stream));
  if (src->stream != dst->stream) {
    HIPCHK(dst, hipEventRecord(dst->ev1, dst->stream));
    HIPCHK(dst, hipStreamWaitEvent(src->stream, dst->ev1, 0));
  }
  if (dst->g.wall_lo && dst->g.wall_hi) {  // a full domain: the rows' neighbours' ghost cells may no longer mirror them
    if (id == fF) { dst->f_ghosts_dirty = true; if (dst->fuse_tm == -1) { dst->tm_decided = false; dst->gas_pending = false; } }
    if (id == fMX || id == fMY) dst->alt_dirty = true;
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
    (void)settle_ahead(h);   // (a predictor formed ahead of its step used the old value)
    h->d.sigma = value;
    h->cd.sigma = value;
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    destroy_graphs(h);
    return VOF_OK;
  }
  settle_ghosts(h);
  // schedule knobs (results never change; tools/sweep_rows.py, tools/variant_ab.py and the tests that
  // force a code path use them): sweeps fused per launch, chunk lengths (0 = heuristic), the
  // equal-cost work plan, the general Jacobi form on square cells, the fused full-domain schedule
  int* knob = !strcmp(name, "jacobi_tb") ? &h->tb : !strcmp(name, "jacobi_tb_adapt") ? &h->tb_adapt
            : !strcmp(name, "jacobi_tb_rows") ? &h->tb_rows : !strcmp(name, "jacobi_tb_general") ? &h->tb_general
            : !strcmp(name, "momentum_rows") ? &h->mom_rows : !strcmp(name, "fctx_rows") ? &h->fctx_rows
            : !strcmp(name, "fctx_corr_rows") ? &h->fctx_corr_rows : !strcmp(name, "band_rows") ? &h->band_rows
            : !strcmp(name, "rows_per_wave") ? &h->rows_override : !strcmp(name, "fuse_transport") ? &h->fuse_transport
            : !strcmp(name, "virtual_ghosts") ? &h->virtual_ghosts : !strcmp(name, "buffer_stores") ? &h->buf_stores : !strcmp(name, "overlap_halves") ? &h->halves : !strcmp(name, "batch_steps") ? &h->step_batch[0] : !strcmp(name, "fuse_tm") ? &h->fuse_tm : !strcmp(name, "tm_rows") ? &h->tm_rows : !strcmp(name, "jacobi_pair") ? &h->jpair : !strcmp(name, "jacobi_pair_rows") ? &h->jpair_rows : !strcmp(name, "pair_vec4") ? &h->pair_vec4 : !strcmp(name, "pair_slow10") ? &h->pair_slow10 : !strcmp(name, "solve_pairs") ? &h->solve_pairs : !strcmp(name, "tb_slow10") ? &h->tb_slow10 : !strcmp(name, "tune_period") ? &h->tune_period : nullptr;
  if (knob) {
    *knob = (int)value;
    if (knob == &h->band_rows && *knob < 1) *knob = 1;
    if (knob == &h->step_batch[0]) *knob = *knob < 4 ? 4 : (*knob & ~1);   // an even number of steps (see vof_step)
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
  if (!strcmp(name, "overlap_halves")) { *value = halves_eligible(h, h->step_batch[vof2d_ctx::kTuneBatch]) ? 1.0 : 0.0; return VOF_OK; }   // effective
  if (!strcmp(name, "gas_share")) { *value = h->gas_share; return VOF_OK; }   // share of exact-zero cells of F the batch-form rule saw (-1: not looked yet)
  if (!strcmp(name, "fuse_transport")) {  // 1 if vof_step runs both FCT sweeps as one kernel on this handle
    *value = (h->g.wall_lo && h->g.wall_hi && h->fuse_transport) ? 1.0 : 0.0;
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
    *value = v ? 1 : 0;   // (plan[0] of an active plan carries its geometry: plan_key)
    return VOF_OK;
  }
#ifdef VOF_WAVE_TIMES
  if (!strncmp(name, "dbg_plan_", 9)) {   // diagnostic build: the plan word in memory and the geometry a k_jacobi_pair launch would expect
    unsigned long long v = 0;
    HIPCHK(h, hipMemcpyAsync(&v, h->d_tbmask + 2 * TB_BANDS * (TB_COLS / 64), sizeof(v), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->jpair_active = true;
    const TbPlan tp = L<double>::tb_plan(h, (int)(h->istep & 1));
    h->jpair_active = false;
    *value = !strcmp(name, "dbg_plan_word") ? (int64_t)v : !strcmp(name, "dbg_plan_waves") ? tp.waves : !strcmp(name, "dbg_plan_R") ? tp.R : tp.ntt;
    return VOF_OK;
  }
#endif
  if (!strcmp(name, "pair_launches")) {   // k_jacobi_pair launches replayed from batch graphs
    *value = h->pair_launches;
    return VOF_OK;
  }
  if (!strcmp(name, "tm_chained_batches")) {   // k_tm batches that found the predictor of their first step in place
    *value = h->tm_chained;
    return VOF_OK;
  }
  if (!strcmp(name, "tm_steps")) {   // steps replayed from batch graphs in the k_tm form
    *value = h->tm_steps;
    return VOF_OK;
  }
  if (!strcmp(name, "tm_choice")) {   // -1: not decided (yet, or the knob decides), 0 / 1: the form the rule (fuse_tm = -1) or the timing (-2) chose
    *value = ((tm_auto(h) || tm_by_rule(h)) && h->tm_decided) ? h->tm_choice : -1;
    return VOF_OK;
  }
  if (!strcmp(name, "halves_steps")) {   // steps replayed from batch graphs in the two-chain form (enqueue_steps_halves)
    *value = h->halves_steps;
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
    // A handle whose batch graphs run the k_tm form is profiled in that form: the same launch sequence, eagerly, every
    // launch between its own event pair (k_momentum, K x k_jacobi_pair / 2 K x k_jacobi_tb, K - 1 x k_tm, k_transport).
    const bool tm_form = ((h->fuse_tm > 0 && tm_eligible(h)) || ((tm_auto(h) || tm_by_rule(h)) && h->tm_decided && h->tm_choice == 1)) && !h->tm_broken &&
                         !h->f_ghosts_dirty && !h->uv_ghosts_dirty && step_leaves_ghosts_virtual(h) && nsteps - done >= 2;
    if (tm_form) {
      // (1 + K x (Jacobi launches + 1) launches, each with its own event pair out of the pool)
      const int per_tm_step = 1 + (DISPATCH_B(h, L<double>::jacobi_pair_ok(h), L<float>::jacobi_pair_ok(h)) ? h->d.jacobi_iters / 10 : h->d.jacobi_iters / 5);
      int K = 2;   // the handle's own batch sizes (an even number of steps each), as far as the event pool allows
      for (int b = vof2d_ctx::kStepBatches - 1; b >= 0; --b)
        if (nsteps - done >= batch_steps(h, 1, b) && 1 + batch_steps(h, 1, b) * per_tm_step <= vof2d_ctx::kMaxTimed && batch_steps(h, 1, b) > K) K = batch_steps(h, 1, b);
      if (1 + K * per_tm_step > vof2d_ctx::kMaxTimed) { h->timed = -1; return fail(h, VOF_ESTATE, "a k_tm batch of two steps has more launches than the profiling event pool"); }
      if (!h->ahead) DISPATCH_T(h, enqueue_tm_head<double>(h, (int)((h->istep + 1) & 1)), enqueue_tm_head<float>(h, (int)((h->istep + 1) & 1)));
      h->ahead = true;
      DISPATCH_T(h, enqueue_steps_tm<double>(h, h->istep + 1, K), enqueue_steps_tm<float>(h, h->istep + 1, K));
      h->istep += K;
      h->ghosts_virtual = true;
      batch = K;
    }
    while (!tm_form && done + batch < nsteps && h->timed + per_step <= vof2d_ctx::kMaxTimed) {
      h->istep += 1;
      h->ahead = false;
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
    if (kid == -2) {   // the second half: cycles inside barriers, cycles in all (the pair kernels)
      if (cap != bufcap) return VOF_EINVAL;
      HIPCHK(h, hipMemcpy(out, buf + 2 * (size_t)bufcap, (size_t)cap * 16, hipMemcpyDeviceToHost));
      return VOF_OK;
    }
    HIPCHK(h, hipMemcpy(out, buf, (size_t)cap * 16, hipMemcpyDeviceToHost));
    return VOF_OK;
  }
  if (cap > bufcap) {
    if (buf) (void)hipFree(buf);
    HIPCHK(h, hipMalloc(&buf, (size_t)cap * 32));   // start / end stamps, then (barrier cycles, all cycles) per wave
    bufcap = cap;
  }
  HIPCHK(h, hipMemset(buf, 0, (size_t)bufcap * 32));
  int k = kid;
  HIPCHK(h, hipMemcpyToSymbol(HIP_SYMBOL(vof::vof_wave_times), &buf, sizeof(buf)));
  HIPCHK(h, hipMemcpyToSymbol(HIP_SYMBOL(vof::vof_wave_kid), &k, sizeof(k)));
  HIPCHK(h, hipMemcpyToSymbol(HIP_SYMBOL(vof::vof_wave_cap), &bufcap, sizeof(bufcap)));
  return VOF_OK;
}
// Diagnostic build only (tools/probes/pair_bound.py): `reps` launches of one pair kernel on the handle's current state
// between one event pair -- k_jacobi_pair (which = 0; p, rhs -> pt, no swap) or k_tm (1: y first, 2: x first; F, u*, v*, p ->
// the twin of F, the second u* / v* pair, rhs: all scratch outside a batch) -- in the ablated form `abl` (ABL_* bits,
// kernels/common.h; wrong values, the state the steps run on is not touched).  plan != 0: k_jacobi_pair on the step's work plan.
extern "C" int vof_debug_time_kernel(vof2d_handle h, int32_t which, int32_t abl, int32_t plan, int32_t reps, float* avg_us) {
  if (!h || !avg_us || reps < 1 || h->d.dtype != VOF_F64 || !buffer_stores_ok(h)) return VOF_EINVAL;
  // abl bit 256: every launch timed on its own behind a 268 MB fill of two arrays the kernels do not touch (rho, nu) --
  // the launch finds neither its inputs nor its last outputs in the L2 / MALL, as it does inside a step
  const bool cold = (abl & 256) != 0;
  abl &= 255;
  double sum_ms = 0.0;
  if (!cold) HIPCHK(h, hipEventRecord(h->ev0, h->stream));
  for (int r = 0; r < reps; ++r) {
    if (cold) {
      HIPCHK(h, hipMemsetAsync(h->fld[fRHO], 0, h->field_elems * h->esz, h->stream));
      HIPCHK(h, hipMemsetAsync(h->fld[fNU], 0, h->field_elems * h->esz, h->stream));
      HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    }
#define ABL_CASE(a) case a: if (which == 0) dbg_pair<a>(h, plan); else if (which == 1) dbg_tm<true, a>(h); else dbg_tm<false, a>(h); break;
    switch (abl) {
      ABL_CASE(0) ABL_CASE(1) ABL_CASE(2) ABL_CASE(3) ABL_CASE(4) ABL_CASE(8) ABL_CASE(16) ABL_CASE(32) ABL_CASE(48) ABL_CASE(19) ABL_CASE(35) ABL_CASE(64) ABL_CASE(192)
      default: return fail(h, VOF_EINVAL, "ablation not instantiated");
    }
#undef ABL_CASE
    if (cold) {
      HIPCHK(h, hipEventRecord(h->ev1, h->stream));
      HIPCHK(h, hipEventSynchronize(h->ev1));
      float ms1 = 0.f;
      HIPCHK(h, hipEventElapsedTime(&ms1, h->ev0, h->ev1));
      sum_ms += ms1;
    }
  }
  float ms = 0.f;
  if (!cold) {
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipEventSynchronize(h->ev1));
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
  } else {
    ms = (float)sum_ms;
  }
  *avg_us = 1e3f * ms / (float)reps;
  return ensure_ok(h);
}
// Diagnostic build only (tools/probes/overlap_tail.py): what would it buy to let the NEXT step's k_jacobi_pair run in the slots
// the tail of k_tm leaves empty?  `reps` times [k_tm, k_jacobi_pair] on the handle's current state, timing only (the Jacobi
// launch reads the rhs the k_tm launch beside it is writing: wrong values, the state the steps run on is not touched):
//   mode 0  both on one stream, one after the other (what the step does);
//   mode 1  k_tm on a stream of the highest priority, k_jacobi_pair on one of the lowest, started together: the dispatcher
//           should hand the Jacobi launch's workgroups only the slots k_tm's pending workgroups do not want;
//   mode 2  the same without priorities (two plain streams);
//   mode 3  the priorities the other way round.
extern "C" int vof_debug_time_overlap(vof2d_handle h, int32_t mode, int32_t reps, float* avg_us) {
  if (!h || !avg_us || reps < 1 || h->d.dtype != VOF_F64 || !buffer_stores_ok(h)) return VOF_EINVAL;
  static hipStream_t sa = nullptr, sb = nullptr, sc = nullptr, sd = nullptr;
  static hipEvent_t e0 = nullptr, ea = nullptr, eb = nullptr;
  if (!sa) {
    int least = 0, greatest = 0;
    HIPCHK(h, hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIPCHK(h, hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, greatest));
    HIPCHK(h, hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, least));
    HIPCHK(h, hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    HIPCHK(h, hipStreamCreateWithFlags(&sd, hipStreamNonBlocking));
    HIPCHK(h, hipEventCreateWithFlags(&e0, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&ea, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&eb, hipEventDisableTiming));
    if (getenv("VOF2D_DEBUG")) fprintf(stderr, "[vof2d] stream priorities: least %d, greatest %d\n", least, greatest);
  }
  hipStream_t const st = h->stream;
  hipStream_t const s_tm = mode == 1 ? sa : mode == 3 ? sb : sc, s_j = mode == 1 ? sb : mode == 3 ? sa : sd;
  HIPCHK(h, hipEventRecord(h->ev0, st));
  for (int r = 0; r < reps; ++r) {
    const bool yf = (r & 1) == 0;
    if (mode == 0) {
      if (yf) dbg_tm<true, 0>(h); else dbg_tm<false, 0>(h);
      dbg_pair<0>(h, 0);
      continue;
    }
    HIPCHK(h, hipEventRecord(e0, st));
    HIPCHK(h, hipStreamWaitEvent(s_tm, e0, 0));
    HIPCHK(h, hipStreamWaitEvent(s_j, e0, 0));
    h->stream = s_tm;
    if (yf) dbg_tm<true, 0>(h); else dbg_tm<false, 0>(h);
    h->stream = s_j;
    dbg_pair<0>(h, 0);
    h->stream = st;
    HIPCHK(h, hipEventRecord(ea, s_tm));
    HIPCHK(h, hipEventRecord(eb, s_j));
    HIPCHK(h, hipStreamWaitEvent(st, ea, 0));
    HIPCHK(h, hipStreamWaitEvent(st, eb, 0));
  }
  HIPCHK(h, hipEventRecord(h->ev1, st));
  HIPCHK(h, hipEventSynchronize(h->ev1));
  float ms = 0.f;
  HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *avg_us = 1e3f * ms / (float)reps;
  return ensure_ok(h);
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
// a capture that failed after the fork may leave a communication stream inside the invalidated capture: the eager launches
// that follow need working ones
static bool comm_streams_usable_after_failed_capture(vof2d_ctx* h) {
  for (hipStream_t* st : {&h->cstream}) {
    if (!*st) continue;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(*st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
      (void)hipGetLastError();
      (void)hipStreamDestroy(*st);
      *st = nullptr;
      if (hipStreamCreateWithFlags(st, hipStreamNonBlocking) != hipSuccess) return false;
    }
  }
  return true;
}
static unsigned field_mask_ok(uint32_t mask) { return mask != 0 && (mask & ~127u) == 0; }
int vof_comm_exchange(vof2d_handle h, uint32_t field_mask) {
  if (!h) return VOF_EINVAL;
  if (!h->comm) return fail(h, VOF_ESTATE, "vof_comm_init has not been called");
  if (!field_mask_ok(field_mask)) return fail(h, VOF_EINVAL, "field_mask: VOF_XCHG_F | _U | _V | _P | _US | _VS | _RHS");
  HIPCHK(h, hipSetDevice(h->device));
  settle_ghosts(h);
  int rc = comm_post(h, field_mask);
  return rc ? rc : comm_join(h);
}

// the kernels of mode 5 need: two-column tiles whose lanes are stored or skipped together, square cells or not (k_jacobi_pair
// falls back to two k_jacobi_tb launches), the fused transport and its virtual ghosts
static bool mode5_ok(const vof2d_ctx* h) {
  return h->fuse_transport && h->tb >= 5 && h->d.jacobi_iters % 5 == 0 && h->d.jacobi_iters >= 5 && h->g.nx >= 16;
}
int vof_step_tm_piece(vof2d_handle h, int32_t piece) {
  if (!h || piece < 0 || piece > 2) return VOF_EINVAL;
  if (h->next_phase != 0) return fail(h, VOF_ESTATE, "a phased step (vof_step_phase) is in progress");
  if (!mode5_ok(h)) return fail(h, VOF_ESTATE, "the pair kernels need the fused transport and five-sweep Jacobi launches");
  if (h->f_ghosts_dirty || h->uv_ghosts_dirty) return fail(h, VOF_ESTATE, "the first step after set_init_F / set_field runs through vof_step");
  (void)settle_ahead(h);   // (a full domain that ran chained k_tm batches: its u*, v*, rhs are the last step's from here on)
  if (piece == 0) {
    DISPATCH_T(h, tm5_head<double>(h), tm5_head<float>(h));
  } else if (piece == 1) {
    h->istep += 1;
    DISPATCH_T(h, (tm5_jacobi<double>(h, (int)(h->istep & 1)), tm5_tm<double>(h, h->istep, 0)), (tm5_jacobi<float>(h, (int)(h->istep & 1)), tm5_tm<float>(h, h->istep, 0)));
    swap_F(h);
    swap_S(h);
  } else {
    h->istep += 1;
    const bool y_first = (h->istep % 2 == 0);
    DISPATCH_T(h, (jacobi_n<double>(h, h->d.jacobi_iters, false, -1), transport_part<double>(h, y_first, kAllOwned)),
               (jacobi_n<float>(h, h->d.jacobi_iters, false, -1), transport_part<float>(h, y_first, kAllOwned)));
    swap_F(h);
    if (!h->virtual_ghosts) DISPATCH_T(h, L<double>::set_bc<BC_ALL>(h), L<float>::set_bc<BC_ALL>(h));
  }
  h->ghosts_virtual = h->virtual_ghosts != 0;
  return ensure_ok(h);
}
// n steps of mode 5: see include/vof2d.h.  The middle steps are replayed two per hipGraph launch (all three pairs of
// arrays -- F / twin, u* v* / mx my, p / pt -- are back where they were after two steps); the head and the tail of a
// call, and an odd middle step, are launched eagerly.
static int step_exchange_mode5(vof2d_handle h, int64_t nsteps) {
  int rc;
  if (!mode5_ok(h)) return fail(h, VOF_ESTATE, "overlap mode 5 needs the fused transport and five-sweep Jacobi launches");
  if (nsteps == 0) return VOF_OK;
  if (h->f_ghosts_dirty || h->uv_ghosts_dirty || h->xchg_steps == 0) {
    // the first step after set_init_F / set_field (the reference's intermediate set_BC calls), and the first of a
    // communicator (RCCL connects on first use): a step of mode 1
    if ((rc = vof_step_exchange(h, 1, 1))) return rc;
    if (--nsteps == 0) return VOF_OK;
  }
  const bool want_graph = !(h->d.flags & VOF_FLAG_NO_GRAPH) && h->xchg_graph && h->xchg5_graph;
  // head
  DISPATCH_T(h, tm5_head<double>(h), tm5_head<float>(h));
  if ((rc = comm_post(h, VOF_XCHG_US | VOF_XCHG_VS | VOF_XCHG_RHS))) return rc;
  if ((rc = comm_join(h))) return rc;
  int64_t mid = nsteps - 1;
  auto eager_mid = [&]() -> int {
    h->istep += 1;
    int r2 = VOF_OK;
    DISPATCH_T(h, r2 = enqueue_mid_step5<double>(h), r2 = enqueue_mid_step5<float>(h));
    h->xchg_steps += 1;
    return r2;
  };
  if (mid & 1) { if ((rc = eager_mid())) return rc; mid -= 1; }
  while (mid > 0) {
    const int key = (int)((h->istep + 1) & 1) | ((h->fld[fF] == h->f_home ? 0 : 1) << 1) | ((h->fld[fUS] == h->us_home ? 0 : 1) << 2) | ((h->fld[fP] == h->p_home ? 0 : 1) << 3);
    if (want_graph && h->xchg5_graph && !h->gxchg5[key]) {
      void* keep[NFIELDS];
      memcpy(keep, h->fld, sizeof(keep));
      const int64_t istep0 = h->istep;
      hipGraph_t graph = nullptr;
      hipError_t e = hipStreamBeginCapture(h->stream, hipStreamCaptureModeRelaxed);
      rc = VOF_OK;
      if (e == hipSuccess) {
        for (int k = 0; k < 2 && rc == VOF_OK; ++k) {
          h->istep += 1;
          DISPATCH_T(h, rc = enqueue_mid_step5<double>(h), rc = enqueue_mid_step5<float>(h));
        }
        e = hipStreamEndCapture(h->stream, &graph);
      }
      h->istep = istep0;
      if (e == hipSuccess && rc == VOF_OK && graph) e = hipGraphInstantiate(&h->gxchg5[key], graph, nullptr, nullptr, 0);
      if (graph) (void)hipGraphDestroy(graph);
      memcpy(h->fld, keep, sizeof(keep));
      if (e != hipSuccess || rc != VOF_OK || !h->gxchg5[key]) {
        (void)hipGetLastError();
        h->gxchg5[key] = nullptr;
        h->xchg5_graph = 0;   // eager from here on
        if (!comm_streams_usable_after_failed_capture(h)) return fail(h, VOF_EHIP, "cannot recreate the communication stream after a failed capture");
        if (getenv("VOF2D_DEBUG")) fprintf(stderr, "[vof2d] mode-5 exchange graph capture failed (%s): eager\n", hipGetErrorString(e));
      }
    }
    if (want_graph && h->xchg5_graph && h->gxchg5[key]) {
      HIPCHK(h, hipGraphLaunch(h->gxchg5[key], h->stream));
      h->istep += 2;
      h->xchg_steps += 2;
      h->xchg_graph_steps += 2;
    } else {
      if ((rc = eager_mid())) return rc;
      if ((rc = eager_mid())) return rc;
    }
    mid -= 2;
  }
  // tail
  h->istep += 1;
  DISPATCH_T(h, rc = enqueue_tail_step5<double>(h), rc = enqueue_tail_step5<float>(h));
  if (rc) return rc;
  h->xchg_steps += 1;
  h->ghosts_virtual = h->virtual_ghosts != 0;
  return ensure_ok(h);
}

int vof_step_exchange(vof2d_handle h, int64_t nsteps, int32_t overlap) {
  if (!h || nsteps < 0 || overlap < 0 || overlap > 5 || overlap == 2) return VOF_EINVAL;   // (2 was retired: never worth it)
  if (!h->comm) return fail(h, VOF_ESTATE, "vof_comm_init has not been called");
  if (h->next_phase != 0) return fail(h, VOF_ESTATE, "a phased step (vof_step_phase) is in progress");
  HIPCHK(h, hipSetDevice(h->device));
  (void)settle_ahead(h);   // (see vof_step_tm_piece)
  if (overlap == 5) return step_exchange_mode5(h, nsteps);
  const bool want_graph = !(h->d.flags & VOF_FLAG_NO_GRAPH);
  for (int64_t s = 0; s < nsteps; ++s) {
    // the captured step leaves the ghost cells virtual (if the handle does that at all); every other
    // way through this loop wants them settled first
    const bool captured_path = want_graph && h->xchg_graph && h->xchg_steps > 0 && !h->f_ghosts_dirty && !h->uv_ghosts_dirty;
    const bool virt = captured_path && h->virtual_ghosts;
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
    // Two mode-4 steps per graph launch (a graph launch leaves ~9 us of idle queue behind it, see vof_step):
    // only once both single-step graphs of this handle exist, i.e. this RCCL has shown that it can be
    // captured; two steps return the F / twin pair and the parity to where they were.
    if (captured_path && overlap == 4 && h->xchg_pair && virt && nsteps - s >= 2 && h->gxchg[par][4][ori] &&
        h->gxchg[par ^ 1][4][ori ^ 1]) {
      if (!h->gxchg2[par][ori]) {
        void* keep[NFIELDS];
        memcpy(keep, h->fld, sizeof(keep));
        hipGraph_t graph = nullptr;
        hipError_t e = hipStreamBeginCapture(h->stream, hipStreamCaptureModeRelaxed);
        rc = VOF_OK;
        if (e == hipSuccess) {
          DISPATCH_T(h, rc = enqueue_step_exchange<double>(h, 4), rc = enqueue_step_exchange<float>(h, 4));
          h->istep += 1;
          if (rc == VOF_OK) DISPATCH_T(h, rc = enqueue_step_exchange<double>(h, 4), rc = enqueue_step_exchange<float>(h, 4));
          h->istep -= 1;
          e = hipStreamEndCapture(h->stream, &graph);
        }
        if (e == hipSuccess && rc == VOF_OK && graph) e = hipGraphInstantiate(&h->gxchg2[par][ori], graph, nullptr, nullptr, 0);
        if (graph) (void)hipGraphDestroy(graph);
        memcpy(h->fld, keep, sizeof(keep));
        if (e != hipSuccess || rc != VOF_OK || !h->gxchg2[par][ori]) {
          (void)hipGetLastError();
          h->gxchg2[par][ori] = nullptr;
          h->xchg_pair = 0;   // single-step graphs from here on (they are known to work)
          if (!comm_streams_usable_after_failed_capture(h)) return fail(h, VOF_EHIP, "cannot recreate the communication stream after a failed capture");
          if (getenv("VOF2D_DEBUG")) fprintf(stderr, "[vof2d] two-step exchange graph capture failed (%s): one step per launch\n", hipGetErrorString(e));
        }
      }
      if (h->gxchg2[par][ori]) {
        HIPCHK(h, hipGraphLaunch(h->gxchg2[par][ori], h->stream));
        h->istep += 1;
        s += 1;
        h->xchg_steps += 2;
        h->xchg_graph_steps += 2;
        h->ghosts_virtual = virt;
        continue;
      }
    }
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
          if (!comm_streams_usable_after_failed_capture(h)) return fail(h, VOF_EHIP, "cannot recreate the communication stream after a failed capture");
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
    if (eo == 1 && (rc = comm_post(h, VOF_XCHG_P))) return rc;
    if ((rc = vof_step_phase(h, 1))) return rc;
    if (eo && (rc = comm_post(h, eo == 3 ? (VOF_XCHG_P | VOF_XCHG_U | VOF_XCHG_V) : (VOF_XCHG_U | VOF_XCHG_V)))) return rc;
    if ((rc = vof_step_phase(h, 2))) return rc;
    if ((rc = comm_post(h, eo ? VOF_XCHG_F : (VOF_XCHG_F | VOF_XCHG_U | VOF_XCHG_V | VOF_XCHG_P)))) return rc;
    if ((rc = comm_join(h))) return rc;
    h->xchg_steps += 1;
  }
  return VOF_OK;
}

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
