// runtime/schedule.h -- the per-step launch schedule (2dvof.py:506-528): sweeps, phases, the fused full-domain step, ghost-cell bookkeeping, graph housekeeping
//
// Part of the host-side runtime of libvof2d_hip.so; included (once, in this order) by vof2d_api.hip:
// context.h, launches.h, schedule.h, comm.h, selftest.h.  Everything here has internal linkage.
#pragma once
#include "launches.h"

namespace {

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
// (the halo rows are the neighbours' to send).
template <typename T>
void final_sweep(vof2d_ctx* h, bool along_x) {
  const int lo = h->d.own_lo > h->g.ilo ? h->d.own_lo : h->g.ilo, hi = h->d.own_hi < h->g.ihi ? h->d.own_hi : h->g.ihi;
  if (hi < lo) return;
  if (along_x) L<T>::template fct_x<true, false>(h, lo, hi); else L<T>::template fct_y<true, false>(h, lo, hi);
}

enum TransportPart { kAllOwned = 0, kEdgeBands = 1, kRest = 2 };
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
// knob "solve_pairs": -1 (default) from 4 M stored cells on -- us per sweep, back to back, five per launch / ten per launch:
// 1024^2 3.10 / 3.41 (the solve of BASELINE configs[1]: 0.93 / 1.05 s), 2048^2 5.99 / 5.16, 4096^2 18.3 / 11.6
inline bool solve_pairs_on(const vof2d_ctx* h) {
  return h->solve_pairs > 0 || (h->solve_pairs < 0 && (long)(h->d.row_hi - h->d.row_lo + 1) * h->g.ny >= 4000000L);
}
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
    // (ten sweeps per launch where the pair kernel applies and no work plan is asked for: the residual-terminated solve,
    //  the verbs, the last step of a strip call)
    if (solve_pairs_on(h) && adapt_par < 0 && left >= 10 && L<T>::jacobi_pair_ok(h)) { L<T>::jacobi_pair(h, cur, oth, -1); left -= 10; }
    else if (tb >= 5 && left >= 5) { L<T>::template jacobi_tb<5>(h, cur, oth, adapt_par); left -= 5; }
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
  if (phase == 0) {
    // cal_nu_rho (:513) is folded into its consumers: rho/nu = f(F[i,j]) recomputed per cell;
    // :514, :517 and the (sweep-invariant, BC-independent) rhs of :239-241 in one pass
    L<T>::momentum(h, virt, adapt_par);     // virt: the previous step's set_BC launch was left out (see enqueue_step)
    jacobi_n<T>(h, h->d.jacobi_iters, false, adapt_par);  // :521-522
    if (!lean) L<T>::template set_bc<BC_P | BC_F>(h);  // p part of :525 / :528; F part of :518 (first step)
  } else if (phase == 1) {
    // :524 inside the first sweep of :526
    if (y_first) sweep_y<T, false, true>(h); else sweep_x<T, false, true>(h);
    if (!merge_bc && !lean) L<T>::template set_bc<BC_UV>(h);  // u, v part of :525
  } else {
    final_sweep<T>(h, /*along_x=*/y_first);  // second sweep, :527 fused, on the owned rows
    swap_F(h);
    if (lean) return;                       // the caller's single set_bc<u,v,F,p> follows
    // F part of :528 on the rows this handle produced; a strip's halo rows arrive with the
    // sender's ghost columns (and may be arriving right now)
    if (merge_bc) L<T>::template set_bc<BC_UV | BC_F>(h);
    else L<T>::template set_bc<BC_F>(h, /*own_rows_only=*/true);
  }
}
template <typename T>
void enqueue_step(vof2d_ctx* h, int64_t istep, bool lean = false, bool virt = false) {
  const bool full = h->g.wall_lo && h->g.wall_hi;
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

// K steady-state steps of a full domain (the lean, virtual-ghost schedule of enqueue_step) with every kernel launched
// twice: on rows [1, s] from the upper chain's stream and on rows [s + 1, nx] from the lower chain's.  The boundary s
// moves UP by D rows from one kernel to the next, D >= the rows any kernel reads beyond the rows it produces (k_momentum
// 3 + 1, k_jacobi_tb<5> 5, k_transport 3), so inside one batch
//   * an upper launch reads only what upper launches before it produced (rows <= s_K + D <= s_(K-1)): the upper chain
//     depends on nothing but itself;
//   * a lower launch reads rows >= s_K + 1 - D: output of the previous kernel's lower AND upper launch -- one event;
//   * nothing an upper launch writes (rows <= s_(K+1) <= s_K - D) is still to be read by a lower launch of an earlier
//     kernel (rows >= s_K + 1 - D), and nothing a lower launch writes (rows > s_K) was or will be read by an upper one.
// What it buys: the chip never drains between kernels -- the next kernel's upper half fills the tail of this kernel's
// lower half, and kernels of different shapes (a Jacobi launch beside a transport launch) share the CUs.  Every row is
// produced once, by the same arithmetic: the values do not change.  Both streams are in capture mode when this runs.
constexpr int kHalvesDrift = 8;
// Measured (tools/probes/halves_sweep.py, ms/step off -> on): 4096^2 fp64 dam-break 0.579 -> 0.560 (late) / 0.592 -> 0.562
// (front), bubble 0.694 -> 0.622, 4096^2 fp32 0.367 -> 0.339, 8192^2 2.30 -> 2.24, 3072^2 0.353 -> 0.344, 2560^2 0.261 -> 0.251;
// 2048^2 fp64 0.172 -> 0.190 and 1024^2 0.087 -> 0.097 (half launches too small to fill the chip): on from 6 M cells and 1024 rows per chain.
inline void swap_S(vof2d_ctx* h) {
  void* t = h->fld[fUS]; h->fld[fUS] = h->fld[fMX]; h->fld[fMX] = t;
  t = h->fld[fVS]; h->fld[fVS] = h->fld[fMY]; h->fld[fMY] = t;
}
// ... and rhs with the (otherwise verb-only) kappa array: the chained k_tm batches, whose every launch leaves the
// previous step's u*, v*, rhs intact beside the next step's (enqueue_steps_tm)
inline void swap_SR(vof2d_ctx* h) {
  swap_S(h);
  void* t = h->fld[fRHS]; h->fld[fRHS] = h->fld[fKAPPA]; h->fld[fKAPPA] = t;
}
inline bool tm_eligible(const vof2d_ctx* h) {
  return h->fuse_tm != 0 && h->g.wall_lo && h->g.wall_hi && h->fuse_transport && h->tb >= 5 && h->d.jacobi_iters % 5 == 0 &&
         h->d.jacobi_iters / 5 % 2 == 0 && h->g.nx >= 16;
}
// Where k_tm has a chance at all -- large fp64 grids (tools/probes/halves_sweep.py, ms/step one chain / chains / k_tm, round 4:
// 4096^2 dam-break 0.601 / 0.576 / 0.547, 8192^2 2.27 / 2.31 / 1.93, but 4096^2 rising bubble 0.682 / 0.612 / 0.808 -- mostly
// liquid --, 4096^2 fp32 0.364 / 0.337 / 0.361, 3072^2 0.351 / 0.346 / 0.346, 2048^2 0.168 / - / 0.188) -- the form is chosen by
// a rule on the state (fuse_tm = -1, the default: decide_batch_form_by_rule) or, for exploration, by timing both (fuse_tm = -2).
constexpr double kTmGasShare = 0.5;   // k_tm from this share of exact-zero cells of F on
// ... and on grids of kTmAlwaysCells and more whatever they hold (runtime/launches.h); both precisions
// ... and in fp32 where the grid is too small for the chains (below 6 M cells the alternative is the plain one-chain sequence, and an
// fp32 pair needs half the LDS of an fp64 one: all pairs of a 2048^2 launch are resident at once) -- rising bubble fp32 2048^2
// (BASELINE configs[4]) 0.146 against 0.161 ms/step; in fp64 the same grid loses 10 % to the pairs (profiles/r06_forms_sweep.txt)
inline bool chains_by_size(const vof2d_ctx* h) { return (long)h->g.nx * h->g.ny >= 6000000L && h->g.nx >= 2 * 1024; }
inline int tm_choice_by_rule(const vof2d_ctx* h, double gas_share) {
  const long cells = (long)(h->g.ihi - h->g.ilo + 1) * h->g.ny;
  return (gas_share >= kTmGasShare || cells >= kTmAlwaysCells || (h->d.dtype == VOF_F32 && !chains_by_size(h))) ? 1 : 0;
}
inline bool tm_size_ok(const vof2d_ctx* h) {
  const long cells = (long)h->g.nx * h->g.ny;
  // (fp64 dam-break, k_tm + k_jacobi_pair against the plain sequence, ms/step: 1024^2 0.131 / 0.087, 1536^2 0.143 / 0.124, 2048^2 0.160 / 0.171,
  //  2560^2 0.217 / 0.240, 3072^2 0.269 / 0.339: from 4 M cells on)
  //  fp32: 2048^2 0.121-0.133 / 0.123-0.142, 2560^2 0.142-0.157 / 0.171-0.176, 3072^2 0.188 / 0.23, 4096^2 0.268 / 0.343)
  return tm_eligible(h) && cells >= 4000000L && h->g.nx >= 2048;
}
inline bool tm_by_rule(const vof2d_ctx* h) { return h->fuse_tm == -1 && tm_size_ok(h); }
inline bool tm_auto(const vof2d_ctx* h) { return h->fuse_tm == -2 && tm_size_ok(h); }
// chains: two; three from 32 M cells (8192^2: 2.30 ms/step in one chain, 2.29 in two, 2.16 in three, 2.17 in four; 4096^2: 0.595 / 0.574 /
// 0.566 / 0.579 inside the front, 0.582 / 0.562 / 0.565 / 0.583 behind it); knob values >= 2 force a count
inline int halves_chains(const vof2d_ctx* h) {
  if (h->halves >= 2) return h->halves > 8 ? 8 : h->halves;
  return (long)h->g.nx * h->g.ny >= 32000000L && h->g.nx >= 3 * 1024 ? 3 : 2;
}
inline bool halves_eligible(const vof2d_ctx* h, int K) {
  const int nj = h->d.jacobi_iters / 5, total = K * (2 + nj);
  // (a chain of few rows is short chunks and little else: 1024 x 8192 in two chains of 512 rows 0.319 -> 0.334 ms/step,
  // 2048 x 8192 0.583 -> 0.562, 8192 x 2048 0.604 -> 0.593)
  const bool wanted = h->halves > 0 || (h->halves < 0 && chains_by_size(h));
  return wanted && h->g.wall_lo && h->g.wall_hi && h->fuse_transport && h->tb >= 5 &&
         h->d.jacobi_iters % 10 == 0 && h->g.nx / halves_chains(h) - (total * kHalvesDrift + 1) / 2 >= 64;
}
// (called outside capture mode: streams and events the capture will need)
inline bool halves_prepare(vof2d_ctx* h, int K) {
  const int P = halves_chains(h), nj = h->d.jacobi_iters / 5, total = K * (2 + nj);
  while ((int)h->chain_streams.size() < P - 1) {
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return false;
    h->chain_streams.push_back(st);
  }
  while ((int)h->hev.size() < (P - 1) * total + K + 1 + (P - 1)) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
    h->hev.push_back(e);
  }
  return true;
}
template <typename T>
bool enqueue_steps_halves(vof2d_ctx* h, int64_t first_step, int K) {
  const int P = halves_chains(h), nj = h->d.jacobi_iters / 5, total = K * (2 + nj), nx = h->g.nx;
  hipStream_t st[8];
  st[0] = h->stream;
  for (int p = 1; p < P; ++p) st[p] = h->chain_streams[p - 1];
  hipEvent_t* const ev_done = h->hev.data();                        // [p * total + n]: launch n of chain p (p < P - 1) is enqueued behind ...
  hipEvent_t* const ev_plan = h->hev.data() + (P - 1) * total;      // [k]: the last chain is through the Jacobi launches of step k
  hipEvent_t const ev_fork = h->hev[(P - 1) * total + K];
  hipEvent_t* const ev_join = h->hev.data() + (P - 1) * total + K + 1;   // [p - 1]
  bool ok = hipEventRecord(ev_fork, st[0]) == hipSuccess;
  for (int p = 1; p < P; ++p) ok = ok && hipStreamWaitEvent(st[p], ev_fork, 0) == hipSuccess;
  int s[8];                                                         // chain p produces rows (s[p - 1], s[p]]
  for (int p = 0; p < P - 1; ++p) s[p] = (int)((long)(p + 1) * nx / P) + (total * kHalvesDrift) / 2;
  int n = 0, n_lastjac = 0;
  auto all = [&](auto&& fn) {
    for (int p = 0; p < P; ++p) {
      if (p > 0 && n > 0) ok = ok && hipStreamWaitEvent(st[p], ev_done[(p - 1) * total + n - 1], 0) == hipSuccess;
      h->stream = st[p];
      fn(p == 0 ? 1 : s[p - 1] + 1, p == P - 1 ? nx : s[p], p == 0);
      if (p < P - 1) ok = ok && hipEventRecord(ev_done[p * total + n], st[p]) == hipSuccess;
    }
    h->stream = st[0];
    for (int p = 0; p < P - 1; ++p) s[p] -= kHalvesDrift;
    ++n;
  };
  // (k_tm inside the chains was measured and is slower than either alone -- 4096^2: 0.658 ms/step against 0.543 for
  // k_tm in one chain and 0.563 for chains of the four kernels: a pair's launch wants the whole chip)
  for (int k = 0; k < K && ok; ++k) {
    const int64_t istep = first_step + k;
    const int par = (int)(istep & 1);
    // The first chain's launch of k_momentum carries the planner block of the step's Jacobi launches (tb_make_plan):
    // it overwrites the plan the previous step's launches read and reads the hit masks they reported, so it waits
    // for EVERY other chain to be through them -- the edges that point up the chains (chain p < P - 1: the event behind
    // its last Jacobi launch; the last chain: ev_plan) ...
    if (k > 0) {
      for (int p = 1; p < P - 1; ++p) ok = ok && hipStreamWaitEvent(st[0], ev_done[p * total + n_lastjac], 0) == hipSuccess;
      ok = ok && hipStreamWaitEvent(st[0], ev_plan[k - 1], 0) == hipSuccess;
    }
    const int n_mom = n;
    all([&](int a, int b, bool first) { L<T>::momentum(h, true, first ? par : -1, a, b); });
    // ... and no chain reads the plan before the planner of ITS step has written it: chain 1's first Jacobi launch is
    // ordered behind chain 0's k_momentum by the chain edge, chains 2 .. only behind chain 1's k_momentum
    for (int p = 2; p < P; ++p) ok = ok && hipStreamWaitEvent(st[p], ev_done[n_mom], 0) == hipSuccess;
    int cur = fP, oth = fPT;
    for (int j = 0; j < nj; ++j) {
      all([&](int a, int b, bool) { L<T>::template jacobi_tb<5>(h, cur, oth, par, a, b); });
      const int t = cur; cur = oth; oth = t;
    }
    n_lastjac = n - 1;
    ok = ok && hipEventRecord(ev_plan[k], st[P - 1]) == hipSuccess;
    all([&](int a, int b, bool) {
      const RowRanges rr{{a, 1, 1}, {b, 0, 0}, {L<T>::transport_rows(h), 1, 1}};
      if (istep % 2 == 0) L<T>::template transport<true>(h, &rr); else L<T>::template transport<false>(h, &rr);
    });
    swap_F(h);
  }
  for (int p = 1; p < P; ++p)
    ok = ok && hipEventRecord(ev_join[p - 1], st[p]) == hipSuccess && hipStreamWaitEvent(st[0], ev_join[p - 1], 0) == hipSuccess;
  return ok;
}

// K steady-state steps of a full domain with the step boundary fused away, and the batch boundary too: the handle is
// AHEAD when this runs -- u*, v*, rhs hold the predictor of the batch's first step (the last k_tm of the batch before, or
// one k_momentum launch in front of the first batch: vof_step) -- so a batch is K x (the step's Jacobi launches, k_tm =
// this step's transport + the next step's momentum), the last k_tm also storing u and v (nothing inside a batch reads
// them from memory; everything outside does).  Every k_tm writes the next predictor into the OTHER set of (u*, v*, rhs)
// arrays -- (mx, my, kappa), verb-only otherwise -- and the host's view of the sets alternates: after an even number of
// steps the view is where it was, holding step K + 1's predictor, and the other set still holds step K's -- what the
// reference's u_star, v_star, rhs hold after K steps (settle_ahead copies it over when somebody asks).
inline void swap_P(vof2d_ctx* h) { void* t = h->fld[fP]; h->fld[fP] = h->fld[fPT]; h->fld[fPT] = t; }
// the step's Jacobi sweeps inside a batch: pairs of five-sweep launches as k_jacobi_pair where the handle allows (each
// leaves its result in the other array of the p / pt pair: the host's view is swapped along, an even number of times
// per batch of an even number of steps), else k_jacobi_tb launches ending in fld[fP]
template <typename T>
void batch_jacobi(vof2d_ctx* h, int par) {
  const int nj = h->d.jacobi_iters / 5;
  if (L<T>::jacobi_pair_ok(h)) {
    for (int j = 0; j < nj / 2; ++j) {
      L<T>::jacobi_pair(h, fP, fPT, par);
      swap_P(h);
    }
    return;
  }
  int cur = fP, oth = fPT;
  for (int j = 0; j < nj; ++j) {
    L<T>::template jacobi_tb<5>(h, cur, oth, par);
    const int t = cur; cur = oth; oth = t;
  }
}
template <typename T>
void enqueue_steps_tm(vof2d_ctx* h, int64_t first_step, int K) {
  h->jpair_active = L<T>::jacobi_pair_ok(h);
  h->jpair_captured = h->jpair_active;
  h->tm_rhs_alt = true;
  for (int k = 0; k < K; ++k) {
    const int64_t istep = first_step + k;
    const int par = (int)(istep & 1);
    batch_jacobi<T>(h, par);
    if (k < K - 1) {
      if (istep % 2 == 0) L<T>::template tm<true, false>(h, par ^ 1); else L<T>::template tm<false, false>(h, par ^ 1);
    } else {
      if (istep % 2 == 0) L<T>::template tm<true, true>(h, par ^ 1); else L<T>::template tm<false, true>(h, par ^ 1);
    }
    swap_SR(h);
    swap_F(h);
  }
  h->tm_rhs_alt = false;
  if (h->jpair_active && ((K * (h->d.jacobi_iters / 10)) & 1)) swap_P(h);   // (never: K is even)
  h->jpair_active = false;
  // (K swaps, an even number: the host's view of the pairs is back where it was)
}
// the one k_momentum launch in front of the first batch of a chain (its planner block plans the geometry of the Jacobi
// kernel the batch will run)
template <typename T>
void enqueue_tm_head(vof2d_ctx* h, int par) {
  h->jpair_active = L<T>::jacobi_pair_ok(h);
  L<T>::momentum(h, true, par);
  h->jpair_active = false;
}
// The handle leaves the chained k_tm batches (a field is read or written from outside, a verb, a parameter): the
// reference's u_star, v_star, rhs after the last step are in the other set of arrays -- copy them into the host's view.
int settle_ahead(vof2d_ctx* h) {
  if (!h->ahead) return VOF_OK;
  h->ahead = false;
  // exactly the cells the predictor writes (:206-233, :238-243) -- u* on i in [2, nx], v* on j in [2, ny], rhs on the
  // interior: everything else of the other set is whatever a verb left in mx, my, kappa
  auto copy = [&](int dst, int src, int i0, int j0) -> hipError_t {
    const size_t pitch = (size_t)h->g.pitch * h->esz;
    const size_t off = ((size_t)(i0 - h->d.row_lo) * h->g.pitch + h->g.col0 + j0) * h->esz;
    return hipMemcpy2DAsync(reinterpret_cast<char*>(h->fld[dst]) + off, pitch, reinterpret_cast<const char*>(h->fld[src]) + off, pitch,
                            (size_t)(h->g.ny - j0 + 1) * h->esz, (size_t)(h->g.nx - i0 + 1), hipMemcpyDeviceToDevice, h->stream);
  };
  HIPCHK(h, copy(fUS, fMX, 2, 1));
  HIPCHK(h, copy(fVS, fMY, 1, 2));
  HIPCHK(h, copy(fRHS, fKAPPA, 1, 1));
  return VOF_OK;
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
#define DISPATCH_B(h, expr_d, expr_f) ((h)->d.dtype == VOF_F64 ? (expr_d) : (expr_f))

// true if the next vof_step runs the fused full-domain schedule (k_momentum, 2 x k_jacobi_tb,
// k_transport) that leaves the ghost cells virtual
bool step_leaves_ghosts_virtual(const vof2d_ctx* h) {
  return h->g.wall_lo && h->g.wall_hi && h->fuse_transport &&
         h->virtual_ghosts && !h->f_ghosts_dirty && !h->uv_ghosts_dirty;
}
// Every entry point that reads or writes fields other than through the fused step calls this
// first: if the last step skipped its set_BC launch, run it now (u, v, F with its twin, p).
void settle_ghosts(vof2d_ctx* h) {
  (void)settle_ahead(h);
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
  for (int b = 0; b < vof2d_ctx::kStepBatches; ++b)
    for (int k = 0; k < 2; ++k)
      for (int o = 0; o < 2; ++o)
        if (h->gbatch[b][k][o]) { (void)hipGraphExecDestroy(h->gbatch[b][k][o]); h->gbatch[b][k][o] = nullptr; }
  for (int b = 0; b < vof2d_ctx::kStepBatches; ++b)
    for (int k = 0; k < 2; ++k)
      for (int o = 0; o < 2; ++o)
        if (h->gbatch_tm[b][k][o]) { (void)hipGraphExecDestroy(h->gbatch_tm[b][k][o]); h->gbatch_tm[b][k][o] = nullptr; }
  h->tune_n = 0; h->tune_age = 0; h->tm_decided = false; h->gas_pending = false; h->tm_broken = false; h->tune_ms[0] = h->tune_ms[1] = 0.f;   // (a changed knob changes what is being compared)
  for (int b = 0; b < vof2d_ctx::kStepBatches; ++b) h->halves_captured[b] = false;
  h->batching = true;   // (a parameter change may be what a capture tripped over: try again)
  for (int k = 0; k < 5; ++k)
    if (h->gphase[k]) { (void)hipGraphExecDestroy(h->gphase[k]); h->gphase[k] = nullptr; }
}

}  // namespace
