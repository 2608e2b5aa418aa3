/*
 * vof2d.h -- C ABI of libvof2d_hip.so, the MI355X (gfx950) drop-in for the hot
 * path of houkensjtu/taichi-2d-vof's 2dvof.py.
 *
 * The reference has no FFI: its boundary is the set of zero-argument
 * @ti.kernel callables (2dvof.py:137,162,198,206,236,269,283,321,385,452),
 * the module-global ti.field objects (2dvof.py:53-93) with
 * .to_numpy()/.from_numpy() (2dvof.py:44,46,535,565) and the 0-D field
 * sigma[None] (2dvof.py:28-29).  Each entry point below names the reference
 * line range it replaces.  Plain pointers and sizes only; every function
 * returns 0 on success or a negative VOF_E* code (message via
 * vof_last_error).  No exceptions or aborts cross this boundary.
 *
 * Layout at the boundary (vof_get_field/vof_set_field): C-contiguous
 * (row_hi-row_lo+1, ny+2) arrays of the handle's dtype, index [i - row_lo][j],
 * ghosts included -- exactly what F.to_numpy() returns in the reference for
 * the full domain (row_lo = 0, row_hi = nx+1).  The internal device pitch and
 * padding are private (vof_field_view exposes them for zero-copy halo
 * exchange).
 *
 * Threading: one host thread per handle; verbs enqueue asynchronously on the
 * handle's HIP stream; vof_get_field, vof_get_counter, vof_sync and the
 * timer functions synchronise that stream.
 */
#ifndef VOF2D_H
#define VOF2D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: VOF_HALO_ROWS grew from jacobi_iters + 6 to jacobi_iters + 8 (overlap mode 5: the momentum march of k_tm reads F, u*, v* one
 * stencil further than the four-kernel step): strips and checkpoints laid out with the old macro are refused by vof_create instead of
 * failing later in vof_comm_init. */
#define VOF_ABI_VERSION 2

/* dtype of every field (2dvof.py:9 default_fp) */
#define VOF_F64 0
#define VOF_F32 1

/* error codes */
#define VOF_OK 0
#define VOF_EINVAL (-1)  /* bad argument / unknown name / size mismatch */
#define VOF_EHIP (-2)    /* a HIP runtime call failed (see vof_last_error) */
#define VOF_ENOMEM (-3)
#define VOF_ESTATE (-4)  /* call not valid in the handle's current state */

/* width (rows) of the per-step deep halo a strip needs on each interior side.
 * Dependency radius of one step in i: modes 0-4 (the four-kernel step) normals 1 +
 * curvature 1 + predictor 1 + n Jacobi sweeps + update_uv 1 + fct_x_sweep 2 (velocity) = n + 5 on the low side,
 * rhs 1 + n + fct_x_sweep 3 + ... = n + 5 on the high side; overlap mode 5 (the pair kernels: the momentum march of
 * k_tm reads F'', u*, v* one stencil further) n + 7; one spare row is allocated.  ABI version 1 had n + 6. */
#define VOF_HALO_ROWS(jacobi_iters) ((jacobi_iters) + 8)

typedef struct vof2d_desc {
  int32_t abi_version;    /* VOF_ABI_VERSION */
  int32_t nx, ny;         /* global interior cells (2dvof.py:19-20) */
  int32_t dtype;          /* VOF_F64 | VOF_F32 (2dvof.py:9) */
  int32_t coord_cast_f32; /* 1: keep .astype(np.float32) of 2dvof.py:43,45 */
  int32_t row_lo, row_hi; /* global rows stored by this handle, inclusive.
                             Full domain: 0 .. nx+1.  A strip stores its owned
                             rows plus halo rows. */
  int32_t own_lo, own_hi; /* rows this handle owns (counters, residuals) */
  int32_t jacobi_iters;   /* sweeps per step; 10 in 2dvof.py:521 */
  int32_t device;         /* HIP device ordinal, -1 = current */
  int32_t flags;          /* VOF_FLAG_* */
  double Lx, Ly;          /* 2dvof.py:22-23 */
  double rho_l, rho_g;    /* :24-25 */
  double nu_l, nu_g;      /* :26-27 */
  double sigma;           /* :28-29 (runtime, see vof_set_param) */
  double gx, gy;          /* :30-31 */
  double dt;              /* :33 */
} vof2d_desc;

#define VOF_FLAG_NO_GRAPH 1 /* vof_step launches kernels eagerly (no hipGraph) */

typedef struct vof2d_ctx* vof2d_handle;

/* Fill *d with the constants of 2dvof.py:19-33 for an nx x ny full domain. */
int vof_desc_default(vof2d_desc* d, int32_t nx, int32_t ny, int32_t dtype);

/* Allocate the fields of 2dvof.py:53-89 that are state or scratch of the hot
 * path, zero-initialised (Taichi zero-fills; the solver relies on it,
 * SURVEY 8c-S5).  stream: a hipStream_t to enqueue on, or NULL to create one. */
int vof_create(const vof2d_desc* d, void* stream, vof2d_handle* out);
int vof_destroy(vof2d_handle h);

/* ---- the reference verbs (zero-argument @ti.kernel callables) ---- */
int vof_set_init_F(vof2d_handle h, int32_t ic);     /* 2dvof.py:137-159 (+102-134) */
int vof_set_BC(vof2d_handle h);                     /* :162-189 */
int vof_cal_nu_rho(vof2d_handle h);                 /* :198-203 */
int vof_get_normal_young(vof2d_handle h);           /* :283-309 */
int vof_advect_upwind(vof2d_handle h);              /* :206-233 */
/* n calls of solve_p_jacobi() (:236-266): rhs once, n Jacobi sweeps
 * (ping-pong p/pt), result in p. */
int vof_solve_p_jacobi(vof2d_handle h, int32_t n);
int vof_update_uv(vof2d_handle h);                  /* :269-280 */
int vof_fct_x_sweep(vof2d_handle h);                /* :321-382 */
int vof_fct_y_sweep(vof2d_handle h);                /* :385-448 */
/* :312-318, sweep order from istep parity (even: y then x) */
int vof_solve_VOF_rudman(vof2d_handle h, int64_t istep);
int vof_post_process_f(vof2d_handle h);             /* :452-455 */

/* nsteps iterations of the solver part of the main loop, 2dvof.py:506-528
 * (istep += 1 first; 10 Jacobi sweeps; x/y sweep alternation), using the fused
 * kernel schedule: on a full domain four launches per step -- k_momentum, two
 * five-sweep k_jacobi_tb, k_transport -- replayed from a hipGraph (the first step after set_init_F /
 * set_field / a single verb runs eagerly with the intermediate boundary launches the reference's
 * :518 / :525 stand for; steady-state steps are replayed in batches of 16 / 8 / 2 per graph launch, as chains of
 * launches on row blocks); large grids (either precision: from 4 M cells on where at least half of the cells are
 * gas, from 16 M cells on whatever they hold, in fp32 also below 6 M -- a rule on the state, looked at once per handle and initial state) run two launches per step --
 * k_jacobi_pair (ten sweeps) and k_tm (the step's transport + the next step's momentum), DESIGN.md 3.5 / 3.6 -- in
 * batches of 32 / 8 / 2 that chain.  Whatever the form, F u v p u_star v_star rhs read back after n steps are the
 * reference's after n steps.  rho / nu / mx / my / kappa scratch is not materialised (the k_tm form uses mx, my,
 * kappa as its second set of u_star, v_star, rhs arrays). */
int vof_step(vof2d_handle h, int64_t nsteps);
/* The same step split at the points where a field becomes final, for drivers that overlap the
 * halo exchange with compute (vof2d/strips.py, vof_step_exchange):
 *   phase 0 = predictor + pressure solve + set_BC(p, F)                      -> p final
 *   phase 1 = update_uv folded into the first FCT sweep + set_BC(u, v)       -> u, v final
 *   phase 2 = second FCT sweep + post_process_f + set_BC(F) on the owned rows -> F final
 * Order 0, 1, 2; phase 0 increments istep.  Halo rows of F are not produced by the second sweep: they
 * are the neighbours' to send. */
int vof_step_phase(vof2d_handle h, int32_t phase);
int vof_get_istep(vof2d_handle h, int64_t* istep);
int vof_set_istep(vof2d_handle h, int64_t istep);

/* Extension (not in the reference, whose :521-522 runs a fixed 10 sweeps; SURVEY 8f-1, BASELINE
 * configs[1] "Jacobi Poisson to 1e-6 residual"): rhs once, then Jacobi sweeps (:258-266) until the
 * residual of the last sweep of a batch is <= tol, checked every check_every sweeps, at most
 * max_iters sweeps.  The sweeps of a batch run fused (five per launch; ten -- k_jacobi_pair -- on square cells from 4 M
 * cells on) and the last launch reduces the two norms itself.
 *   VOF_RESID_ABS:  residual = max|p_new - p|                                  over owned rows
 *   VOF_RESID_REL:  residual = max|p_new - p| / max(max|p_new|, VOF_RESID_TINY)   
 * A non-finite update (a diverged field) reads as residual = +inf and ends the solve; that is the only
 * way to +inf: a finite update over a tiny max|p_new| gives a relative residual clamped to DBL_MAX.
 * *residual is this handle's local value; a multi-GPU driver all-reduces the two norms of
 * vof_jacobi_sweeps_norms itself (vof2d/strips.py). */
#define VOF_RESID_ABS 0
#define VOF_RESID_REL 1
#define VOF_RESID_TINY 1e-300
int vof_solve_p(vof2d_handle h, double tol, int32_t max_iters, int32_t check_every, int32_t criterion,
                int32_t* iters_done, double* residual);
/* = vof_solve_p(..., VOF_RESID_ABS, ...) */
int vof_solve_p_residual(vof2d_handle h, double tol, int32_t max_iters, int32_t check_every,
                         int32_t* iters_done, double* residual);
/* rhs build (if build_rhs) + n sweeps; *max_update = max|p_new - p| and *max_p = max|p_new| of the
 * last sweep over this handle's owned rows (a NaN entry counts as +inf). */
int vof_jacobi_sweeps_norms(vof2d_handle h, int32_t n, int32_t build_rhs, double* max_update, double* max_p);
/* the max_update half of vof_jacobi_sweeps_norms */
int vof_jacobi_sweeps_residual(vof2d_handle h, int32_t n, int32_t build_rhs, double* residual);
/* the residual of a criterion from the two (possibly all-reduced) norms; +inf for a non-finite update */
double vof_residual_value(double max_update, double max_p, int32_t criterion);

/* ---- fields: F.to_numpy() / F.from_numpy() (2dvof.py:44,46,535,565) ----
 * names: F u v p u_star v_star mx my kappa rho nu rhs */
int vof_get_field(vof2d_handle h, const char* name, void* dst, size_t nbytes);
int vof_set_field(vof2d_handle h, const char* name, const void* src, size_t nbytes);
/* rows [g0, g1] (global indices, inclusive) of a field, dense (g1-g0+1, ny+2) */
int vof_get_rows(vof2d_handle h, const char* name, int32_t g0, int32_t g1, void* dst, size_t nbytes);
int vof_set_rows(vof2d_handle h, const char* name, int32_t g0, int32_t g1, const void* src,
                 size_t nbytes);
/* device view for zero-copy halo exchange: element (i, j) lives at
 * base + ((i - row_lo) * pitch + col0 + j) * elem_size */
int vof_field_view(vof2d_handle h, const char* name, void** base, int64_t* pitch, int64_t* col0,
                   int64_t* nrows);
/* device-to-device copy of rows [g0, g1] of `name` from src into dst (same
 * nx, ny, dtype; rows must be stored by both).  Used for single-GPU strip
 * emulation; across GPUs the exchange is RCCL send/recv on vof_field_view.
 * Asynchronous and ordered on both handles: the copy follows everything enqueued
 * on src so far, and everything enqueued on src or dst afterwards follows the copy. */
int vof_copy_rows(vof2d_handle dst, vof2d_handle src, const char* name, int32_t g0, int32_t g1);

/* ---- display fields (2dvof.py:458-492; full-domain handles only) ----
 * which: "vof" get_vof_field :458-462, "u" get_u_field :465-470, "v" get_v_field :473-478,
 * "vnorm" get_vnorm_field :481-486.  dst: the rgb_buf image, dense (2*nx, 2*ny) of the field dtype. */
int vof_get_vis_field(vof2d_handle h, const char* which, void* dst, size_t nbytes);
/* interp_velocity :488-492: cell-centred velocity V, dense (nx+2, ny+2, 2) of the field dtype. */
int vof_interp_velocity(vof2d_handle h, void* dst, size_t nbytes);

/* ---- scalars ---- */
/* settable: sigma (sigma[None], :28-29).  readable: sigma dt dx dy dxi dyi
 * dxi2 dyi2 Lx Ly rho_l rho_g nu_l nu_g gx gy (Python-double values).
 * Schedule knobs (settable, results never change): jacobi_tb (sweeps fused per launch: 5, 2 or 1),
 * jacobi_tb_adapt (the equal-cost work plan of the fused Jacobi launches), jacobi_tb_general (the
 * general-stencil form on square cells too), chunk lengths jacobi_tb_rows, momentum_rows, fctx_rows,
 * fctx_corr_rows, band_rows, rows_per_wave (rows a wave marches; 0 = heuristic), fuse_transport
 * (update_uv + both sweeps as one kernel on full domains; readable: 1 if in effect), virtual_ghosts
 * (no set_BC launch in steady-state steps).  Environment: VOF2D_DEBUG (trace to stderr), VOF2D_RCCL
 * (path of the RCCL to bind), VOF2D_XCHG_GRAPH=0 (never capture the exchange). */
int vof_set_param(vof2d_handle h, const char* name, double value);
int vof_get_param(vof2d_handle h, const char* name, double* value);
/* "courant_violations": number of faces that tripped the prints at
 * 2dvof.py:274-275,279-280 since creation (owned rows only). */
int vof_get_counter(vof2d_handle h, const char* name, int64_t* value);

/* ---- sync / timing / errors ---- */
int vof_sync(vof2d_handle h);
/* hipEvent pair on the handle's stream */
int vof_timer_start(vof2d_handle h);
int vof_timer_stop(vof2d_handle h, float* ms); /* records, synchronises, returns elapsed */
/* Built-in in-situ profiler: nsteps steps of the fused schedule with a start/stop event pair on
 * every dispatch (hipExtLaunchKernelGGL); per-kernel sums accumulate until vof_reset_profile.
 * kernel names: k_momentum k_set_bc k_jacobi k_jacobi_tb k_correct k_fct_x k_fct_y (+ k_normals
 * k_kappa k_predictor k_rhs when the momentum fusion is off).  The state advances by nsteps. */
int vof_profile_steps(vof2d_handle h, int64_t nsteps);
int vof_get_profile(vof2d_handle h, const char* kernel, double* avg_us, int64_t* launches);
int vof_reset_profile(vof2d_handle h);
/* n (even) Jacobi sweeps of the current rhs, back to back, between one hipEvent pair on the
 * handle's stream; *ms_per_sweep = elapsed / n.  p advances by n sweeps.  Uses the kernels the
 * step uses (k_jacobi_tb launches of `jacobi_tb` sweeps, k_jacobi when that parameter is 1). */
int vof_time_jacobi(vof2d_handle h, int32_t n, float* ms_per_sweep);
/* ---- strips over RCCL (extension: the reference is single-device) ----
 * A handle that stores a row strip (row_lo..row_hi with VOF_HALO_ROWS halo rows on each interior
 * side of own_lo..own_hi) can run the per-step halo exchange itself: rank r of `world` owns the
 * r-th strip counted from the left wall and trades W = VOF_HALO_ROWS(jacobi_iters) rows of F, u,
 * v, p with ranks r-1 and r+1 by ncclSend / ncclRecv, straight from / to field memory, on a
 * communication stream of its own.  RCCL is bound at run time (librccl.so.1; a copy already in the
 * process, e.g. PyTorch's, is reused), so single-GPU users need no RCCL.
 *   vof_comm_get_unique_id  on one rank; ship the VOF_COMM_ID_BYTES to the others (any transport)
 *   vof_comm_init           collective over all ranks (ncclCommInitRank)
 *   vof_step_exchange       nsteps x [phase 0, send/recv p, phase 1, send/recv u v, phase 2,
 *                           send/recv F, join]: each field leaves as soon as it is final for the
 *                           step and travels under the remaining kernels (overlap = 1).
 *                           overlap = 0: one exchange of all four after the step; overlap = 3: p, u,
 *                           v in one group after phase 1, F after phase 2 (one fork less); overlap = 4
 *                           (what the drivers use): update_uv and both sweeps as ONE kernel
 *                           (k_transport), first on the edge bands, then
 *                           send/recv p, u, v, F in one group, and the same kernel on the remaining
 *                           rows while they travel.  After the first step (RCCL connects on
 *                           first use) a step and its exchanges are one hipGraph launch (mode 4: two
 *                           steps per launch once both single-step graphs exist); if the
 *                           RCCL at hand cannot be captured the launches stay eager.  The host
 *                           does not block
 *   vof_comm_exchange       one exchange of the fields in field_mask, then join (for verb-level
 *                           drivers, e.g. the residual-terminated pressure solve)
 * VOF_COMM_LOOPBACK (self-test on one GPU): both neighbours are the calling rank itself. */
#define VOF_COMM_ID_BYTES 128
#define VOF_COMM_LOOPBACK 1
#define VOF_XCHG_F 1u
#define VOF_XCHG_U 2u
#define VOF_XCHG_V 4u
#define VOF_XCHG_P 8u
#define VOF_XCHG_US 16u   /* u*, v*, rhs: the exchange state of overlap mode 5 (next to F and p) */
#define VOF_XCHG_VS 32u
#define VOF_XCHG_RHS 64u
int vof_comm_get_unique_id(void* id /* VOF_COMM_ID_BYTES */);
int vof_comm_init(vof2d_handle h, const void* id, int32_t rank, int32_t world, int32_t flags);
int vof_comm_exchange(vof2d_handle h, uint32_t field_mask);
int vof_step_exchange(vof2d_handle h, int64_t nsteps, int32_t overlap);
/* overlap = 5: the strips run the kernels the single GPU runs.  The step boundary moves behind the momentum predictor:
 *   one call of n steps = k_momentum (owned rows), send/recv u*, v*, rhs
 *                         + (n - 1) x [k_jacobi_pair (ten sweeps, all stored rows), k_tm (this step's transport + the next
 *                           step's momentum) on the two edge bands beside k_tm on the other owned rows, send/recv F, u*, v*,
 *                           rhs, p in one group under the latter]
 *                         + [two k_jacobi_tb launches, k_transport on the edge bands, send/recv F, u, v, p, k_transport on
 *                           the other rows]  (the last step is mode 4's: u and v reach memory only there),
 * the middle steps two per hipGraph launch.  Halo depth: ten sweeps + 7 rows of the fused transport / momentum
 * marches <= VOF_HALO_ROWS.  vof_step_tm_piece runs the kernels of one piece WITHOUT the exchange (0: the k_momentum of
 * the first step, 1: one middle step, 2: the last step), for drivers that move the halos themselves (vof_copy_rows between
 * strip handles on one device: tests, tools/predict_strips.py). */
int vof_step_tm_piece(vof2d_handle h, int32_t piece);
/* max over all ranks of *value (ncclAllReduce on the compute stream, then a stream sync): the
 * global residual of the residual-terminated pressure solve, and a barrier for timing loops. */
int vof_comm_allreduce_max(vof2d_handle h, double* value);
/* ncclGetVersion code of the RCCL in use (0: none) and whether vof_step_exchange replays captured
 * graphs with this handle (verified from RCCL 2.27.7 on; older copies launch eagerly). */
int vof_comm_info(vof2d_handle h, int32_t* rccl_version, int32_t* graph_capture);
int vof_comm_destroy(vof2d_handle h);

/* Self-test of the kernels' exact constant-denominator division (the Jacobi update p = num / ap,
 * 2dvof.py:262-264, and the / dx, / dy, / dt, / (dx*dy) of :207-232 and :327-449 are evaluated as
 * Markstein-corrected multiplications by the reciprocal): n generated (numerator, denominator)
 * pairs -- ordinary, tiny, huge, special, and subnormal quotients on and beside the midpoints of
 * the subnormal grid -- are divided by that routine on the current device.  a_out, b_out, q_out:
 * host arrays of n elements of `dtype`; the caller checks q == a / b with the host's IEEE division
 * (tests/test_parity_gpu.py). */
int vof_selftest_division(int32_t dtype, int64_t n, uint64_t seed, void* a_out, void* b_out, void* q_out);
const char* vof_last_error(vof2d_handle h);
/* "hip-gfx950" for the product library, "cpu-oracle" for oracle/ */
const char* vof_backend(void);

#ifdef __cplusplus
}
#endif
#endif /* VOF2D_H */
