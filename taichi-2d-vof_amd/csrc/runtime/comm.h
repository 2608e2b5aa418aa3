// runtime/comm.h -- strips over RCCL: run-time binding (dlopen), the halo send/recv groups, the step with its exchanges
//
// Part of the host-side runtime of libvof2d_hip.so; included (once, in this order) by vof2d_api.hip:
// context.h, launches.h, schedule.h, comm.h, selftest.h.  Everything here has internal linkage.
#pragma once
#include "schedule.h"

namespace {

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
  for (int a = 0; a < 2; ++a)
    for (int o = 0; o < 2; ++o)
      if (h->gxchg2[a][o]) { (void)hipGraphExecDestroy(h->gxchg2[a][o]); h->gxchg2[a][o] = nullptr; }
  for (int k = 0; k < 16; ++k)
    if (h->gxchg5[k]) { (void)hipGraphExecDestroy(h->gxchg5[k]); h->gxchg5[k] = nullptr; }
}
// (s_in_alt: between the launches of k_tm and the host's swap the new u*, v* still live in the mx / my arrays;
// on_cstream: whatever the messages wait for was enqueued on the communication stream itself)
constexpr int kTmBandRows = 6;    // rows per pair chunk of k_tm's edge-band launch
constexpr int kTmReachRows = 8;   // halo rows of F, u*, v* the marches of k_tm read beyond the owned rows (3 + 3 + 1: x pipeline, momentum window, faces)
// shallow: F, u*, v* travel kTmReachRows deep only (mode 5's middle steps: all k_tm reads of them; p and rhs serve the ten
// sweeps in front of it and travel W deep)
int comm_post(vof2d_ctx* h, unsigned mask, bool f_in_twin = false, int fork = -1, bool s_in_alt = false, bool on_cstream = false, bool shallow = false) {
  Rccl* r = rccl();
  const int W = VOF_HALO_ROWS(h->d.jacobi_iters);
  const size_t row_bytes = (size_t)h->g.pitch * h->esz;
  if (!on_cstream) {
    hipEvent_t ready = fork >= 0 ? h->ev_fork[fork] : h->ev_ready;
    HIPCHK(h, hipEventRecord(ready, h->stream));
    HIPCHK(h, hipStreamWaitEvent(h->cstream, ready, 0));
  }
  static const int ids[7] = {fF, fU, fV, fP, fUS, fVS, fRHS};
  NCCLCHK(h, r->GroupStart());
  // Inside the group no early return: a failing send / recv must still be followed by GroupEnd, or
  // the next (eager) exchange would nest inside the group left open and never be issued.
  int first_err = 0;
  const char* what = "";
  auto note = [&](int rc, const char* call) { if (rc != 0 && first_err == 0) { first_err = rc; what = call; } };
  for (int k = 0; k < 7; ++k) {
    if (!(mask & (1u << k))) continue;
    // between the two transport phases the new F still lives in the twin buffer
    const int id = (k == 0 && f_in_twin) ? fF2 : (k == 4 && s_in_alt) ? fMX : (k == 5 && s_in_alt) ? fMY : ids[k];
    char* base = reinterpret_cast<char*>(h->fld[id]);
    auto row = [&](int g) { return base + (size_t)(g - h->d.row_lo) * row_bytes; };
    const int D = (shallow && (k == 0 || k == 4 || k == 5) && kTmReachRows < W) ? kTmReachRows : W;   // rows of this field
    const size_t bytes = (size_t)D * row_bytes;
    if (h->peer_lo >= 0) {
      note(r->Send(row(h->d.own_lo), bytes, /*ncclInt8*/ 0, h->peer_lo, h->comm, h->cstream), "ncclSend(lo)");
      note(r->Recv(row(h->d.own_lo - D), bytes, 0, h->peer_lo, h->comm, h->cstream), "ncclRecv(lo)");
    }
    if (h->peer_hi >= 0) {
      note(r->Send(row(h->d.own_hi - D + 1), bytes, 0, h->peer_hi, h->comm, h->cstream), "ncclSend(hi)");
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


// One step with its exchanges on (compute stream, communication stream).  mode 0: one exchange of
// all four fields after the step; 1: each field leaves as soon as it is final (p after phase 0,
// u, v after phase 1, F after phase 2); 3: p, u, v together after phase 1, F after phase 2 (one fork
// less); 4: the fused transport, edge bands first, one group for all four fields (the default of the
// drivers).  Enqueued eagerly or under stream capture.
template <typename T>
int enqueue_step_exchange(vof2d_ctx* h, int mode) {
  int rc;
  // lean phases (no boundary launch inside): the rows travel with whatever ghost columns they
  // have, and one set_bc<u,v,F,p> over all stored rows -- owned and received alike -- follows the
  // join.  Only reached on steps that start with consistent F ghosts (vof_step_exchange).
  // With virtual ghosts (see enqueue_step) even that launch goes: the rows travel with stale ghost
  // columns and the next step's k_momentum forms the ones it reads, for owned and received rows alike.
  const bool lean = true;
  const bool virt = h->virtual_ghosts != 0;
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
  if (mode == 1 && (rc = comm_post(h, VOF_XCHG_P, false, 0))) return rc;   // p is final
  enqueue_phase<T>(h, 1, h->istep, false, lean);
  if (mode && (rc = comm_post(h, mode == 3 ? (VOF_XCHG_P | VOF_XCHG_U | VOF_XCHG_V) : (VOF_XCHG_U | VOF_XCHG_V), false, 1))) return rc;  // u, v are final
  enqueue_phase<T>(h, 2, h->istep, false, lean);
  if ((rc = comm_post(h, mode ? VOF_XCHG_F : (VOF_XCHG_F | VOF_XCHG_U | VOF_XCHG_V | VOF_XCHG_P), false, 2))) return rc;
  if ((rc = comm_join(h))) return rc;           // halos complete before the next step
  if (lean && !virt) L<T>::template set_bc<BC_ALL>(h);
  return VOF_OK;
}

// ---- overlap mode 5: the strips run the pair kernels (include/vof2d.h, vof_step_exchange) ----
// owned rows of the handle inside the computable rows, and the two W-row bands at its interior edges
struct OwnedRows { int lo, hi, in_lo, in_hi; bool band_lo, band_hi; };
inline OwnedRows owned_rows(const vof2d_ctx* h) {
  const int W = VOF_HALO_ROWS(h->d.jacobi_iters);
  OwnedRows o;
  o.lo = h->d.own_lo > h->g.ilo ? h->d.own_lo : h->g.ilo;
  o.hi = h->d.own_hi < h->g.ihi ? h->d.own_hi : h->g.ihi;
  o.band_lo = !h->g.wall_lo; o.band_hi = !h->g.wall_hi;
  o.in_lo = o.band_lo ? o.lo + W : o.lo;
  o.in_hi = o.band_hi ? o.hi - W : o.hi;
  return o;
}
// the first step's k_momentum: u*, v*, rhs of the owned rows (their halo rows arrive by exchange: mode 5's state)
template <typename T>
void tm5_head(vof2d_ctx* h) {
  // The middle steps alternate u*, v* between their own arrays and mx, my, and a call may end with the host's view on
  // the second pair: the cells the predictor never writes (u* on i = 1, v* on j = 1, ny + 1, the ghost cells) must hold
  // the zeros the reference's never-written entries hold (S5), not what a verb (get_normal_young) left there.
  if (h->alt_dirty) {
    (void)hipMemsetAsync(h->fld[fMX], 0, h->field_elems * h->esz, h->stream);
    (void)hipMemsetAsync(h->fld[fMY], 0, h->field_elems * h->esz, h->stream);
    h->alt_dirty = false;
  }
  const OwnedRows o = owned_rows(h);
  h->jpair_active = L<T>::jacobi_pair_ok(h);    // (the planner block plans the geometry of the kernel that will run)
  L<T>::momentum(h, true, (int)((h->istep + 1) & 1), o.lo, o.hi);
  h->jpair_active = false;
}
// the ten sweeps of a middle step on all stored rows
template <typename T>
void tm5_jacobi(vof2d_ctx* h, int par) {
  if (L<T>::jacobi_pair_ok(h)) {
    // jacobi_iters / 10 launches of ten sweeps each, like batch_jacobi (an odd count leaves the host's view of the p / pt
    // pair swapped: the exchange graphs are keyed by it)
    h->jpair_active = true;
    for (int j = 0; j < h->d.jacobi_iters / 10; ++j) {
      L<T>::jacobi_pair(h, fP, fPT, par);
      swap_P(h);
    }
    h->jpair_active = false;
  } else {
    jacobi_n<T>(h, h->d.jacobi_iters, false, -1);
  }
}
// k_tm on the owned rows: part 0 = all, 1 = the edge bands (short chunks), 2 = the rest (carries the planner block)
template <typename T>
void tm5_tm(vof2d_ctx* h, int64_t istep, int part) {
  const OwnedRows o = owned_rows(h);
  const bool y_first = (istep % 2 == 0);
  const int par_next = (int)((istep + 1) & 1);
  h->jpair_active = L<T>::jacobi_pair_ok(h);
  auto run = [&](int a, int b, int par, int rows_forced, int a2 = 1, int b2 = 0) {
    if (b < a) { a = a2; b = b2; a2 = 1; b2 = 0; }
    if (b < a) return;
    if (y_first) L<T>::template tm<true, false>(h, par, a, b, rows_forced, a2, b2); else L<T>::template tm<false, false>(h, par, a, b, rows_forced, a2, b2);
  };
  const bool split = o.in_lo <= o.in_hi && (o.band_lo || o.band_hi);
  if (part == 0 || !split) {
    if (part == 0 || (part == 1 && (o.band_lo || o.band_hi)) || (part == 2 && !(o.band_lo || o.band_hi))) run(o.lo, o.hi, par_next, 0);   // (one launch for the step: it carries the planner block)
  } else if (part == 1) {
    // both bands in ONE launch, in short chunks (a pair's march is its rows + 14 steps whatever its rows: the bands are
    // what the send / recv group waits for -- 8192-wide interior strip of 8: two launches of 18-row chunks 56 + 69 us)
    run(o.band_lo ? o.lo : 1, o.band_lo ? o.in_lo - 1 : 0, -1, kTmBandRows, o.band_hi ? o.in_hi + 1 : 1, o.band_hi ? o.hi : 0);
  } else {
    run(o.in_lo, o.in_hi, par_next, 0);
  }
  h->jpair_active = false;
}
// one middle step with its exchange: the edge bands on the communication stream in front of the send / recv group,
// the other rows on the compute stream beside them (launches of k_tm on disjoint rows read the old arrays and write
// the new ones: they do not depend on each other)
template <typename T>
int enqueue_mid_step5(vof2d_ctx* h) {
  int rc;
  tm5_jacobi<T>(h, (int)(h->istep & 1));
  HIPCHK(h, hipEventRecord(h->ev_fork[1], h->stream));
  HIPCHK(h, hipStreamWaitEvent(h->cstream, h->ev_fork[1], 0));
  hipStream_t st = h->stream;
  h->stream = h->cstream;
  tm5_tm<T>(h, h->istep, 1);
  h->stream = st;
  if ((rc = comm_post(h, VOF_XCHG_F | VOF_XCHG_US | VOF_XCHG_VS | VOF_XCHG_RHS | VOF_XCHG_P, /*f_in_twin=*/true, 1, /*s_in_alt=*/true, /*on_cstream=*/true, /*shallow=*/true))) return rc;
  tm5_tm<T>(h, h->istep, 2);
  swap_F(h);
  swap_S(h);
  return comm_join(h);
}
// the last step of a call: mode 4's step without its k_momentum (u, v reach memory here)
template <typename T>
int enqueue_tail_step5(vof2d_ctx* h) {
  int rc;
  jacobi_n<T>(h, h->d.jacobi_iters, false, -1);     // (uniform chunks: the plan in memory is of the pairs' geometry)
  const bool y_first = (h->istep % 2 == 0);
  transport_part<T>(h, y_first, kEdgeBands);
  if ((rc = comm_post(h, VOF_XCHG_P | VOF_XCHG_F | VOF_XCHG_U | VOF_XCHG_V, /*f_in_twin=*/true, 1))) return rc;
  transport_part<T>(h, y_first, kRest);
  swap_F(h);
  if ((rc = comm_join(h))) return rc;
  if (!h->virtual_ghosts) L<T>::template set_bc<BC_ALL>(h);
  return VOF_OK;
}

}  // namespace
