/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU oracle: scalar-C restatement of the hot path of
 * /root/reference/2dvof.py (see vof_oracle_impl.inc for the per-function
 * file:line citations).  Exports the same entry points as include/vof2d.h
 * with the prefix `ovof_` so the parity tests can drive oracle and HIP
 * library through one code path.
 *
 * PARITY PIN: taichi==1.4.1 cannot be installed in this image and the
 * reference's test/ scripts hold no golden vectors.  This restatement is
 * pinned to the reference's own source text, executed: tests/golden/ref_*.npz
 * come from /root/reference/2dvof.py run unmodified under a pure-Python
 * stand-in for the taichi module (tests/golden/make_ref_golden.py): 200 x 200
 * as shipped, -ic 1/2/3, 1000 steps, in doubles AND in the shipped precision
 * f32 (Taichi's static typing emulated with NumPy float32 scalars); 128 x 128
 * -ic 1 (BASELINE configs[0]); rectangular-cell runs with only the grid-size
 * literals replaced; runs whose event loop is fed SPACE releases, recording the
 * display buffers (2dvof.py:458-492, :531-559).  tests/test_ref_golden.py
 * requires all 19 arrays of every recorded step -- and rgb_buf / V / the arrow
 * list -- to be reproduced exactly, fp64 and fp32.  NOT pinned: Taichi's code
 * generation (fast_math contraction / reassociation).  Also checked against the
 * independent NumPy restatement (oracle/vof_oracle_np.py) and the
 * self-generated fixtures in tests/golden/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  Build: see oracle/Makefile (-O2 -ffp-contract=off).
 */
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/vof2d.h"

/* constants in Python double, folded exactly as the Python-scope expressions
 * of 2dvof.py */
typedef struct vof_consts {
  double dt, dx, dy, dxi, dyi, dxi2, dyi2, rho_l, rho_g, nu_l, nu_g, sigma, gx, gy;
  double nrm_x, nrm_y, kap_x, kap_y, dxdy, dtdy, dtdx, cfl_x, cfl_y, half_dx, half_dy, sqrt2dx, tiny;
} vof_consts;

/* k-th entry of hstack((0, linspace(0, L, n+1), L)).astype(float32)  (2dvof.py:43,45).
 * np.linspace: arange(n+1) * (L / n), last sample forced to L. */
static double vof_node_coord(double L, int n, int k, int cast_f32) {
  double v;
  if (k == 0)
    v = 0.0;
  else if (k >= n + 1)
    v = L;
  else
    v = (double)(k - 1) * (L / (double)n);
  if (cast_f32) v = (double)(float)v;
  return v;
}

static void vof_consts_compute(const vof2d_desc* d, vof_consts* c) {
  /* dx = x[imin+2] - x[imin+1], Python-scope reads of field values (:47-48).
   * With dtype f32 the field stores float32 in either coord mode. */
  int cast = d->coord_cast_f32 || d->dtype == VOF_F32;
  double dx = vof_node_coord(d->Lx, d->nx, 3, cast) - vof_node_coord(d->Lx, d->nx, 2, cast);
  double dy = vof_node_coord(d->Ly, d->ny, 3, cast) - vof_node_coord(d->Ly, d->ny, 2, cast);
  double dxi = 1 / dx, dyi = 1 / dy;
  c->dt = d->dt; c->dx = dx; c->dy = dy; c->dxi = dxi; c->dyi = dyi;
  c->dxi2 = pow(dxi, 2.0); /* dxi ** 2 */
  c->dyi2 = pow(dyi, 2.0);
  c->rho_l = d->rho_l; c->rho_g = d->rho_g; c->nu_l = d->nu_l; c->nu_g = d->nu_g;
  c->sigma = d->sigma; c->gx = d->gx; c->gy = d->gy;
  c->nrm_x = -1 / (2 * dx);
  c->nrm_y = -1 / (2 * dy);
  c->kap_x = 1 / dx / 2;
  c->kap_y = 1 / dy / 2;
  c->dxdy = dx * dy;
  c->dtdy = d->dt * dy;
  c->dtdx = d->dt * dx;
  c->cfl_x = 0.25 * dx;
  c->cfl_y = 0.25 * dy;
  c->half_dx = dx / 2;
  c->half_dy = dy / 2;
  c->sqrt2dx = sqrt(2.0) * dx;
  c->tiny = 1e-10;
}

/* grids below this many cells run single-threaded (fork/join costs more than the loop) */
#define OMP_MIN_CELLS 100000
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define REAL double
#define SUF(x) CAT(x, _f64)
#define SQRT sqrt
#define FABS fabs
#include "vof_oracle_impl.inc"
#undef REAL
#undef SUF
#undef SQRT
#undef FABS

#define REAL float
#define SUF(x) CAT(x, _f32)
#define SQRT sqrtf
#define FABS fabsf
#include "vof_oracle_impl.inc"
#undef REAL
#undef SUF
#undef SQRT
#undef FABS

struct vof2d_ctx {
  vof2d_desc d;
  vof_consts c;
  grid_f64 g64;
  grid_f32 g32;
  int64_t istep;
  int next_phase;
  char err[256];
};

#define DISPATCH(h, fn, ...)                       \
  do {                                             \
    if ((h)->d.dtype == VOF_F64)                   \
      fn##_f64(&(h)->g64, ##__VA_ARGS__);          \
    else                                           \
      fn##_f32(&(h)->g32, ##__VA_ARGS__);          \
  } while (0)

int ovof_desc_default(vof2d_desc* d, int32_t nx, int32_t ny, int32_t dtype) {
  if (!d || nx < 3 || ny < 3) return VOF_EINVAL;
  memset(d, 0, sizeof(*d));
  d->abi_version = VOF_ABI_VERSION;
  d->nx = nx; d->ny = ny; d->dtype = dtype; d->coord_cast_f32 = 1;
  d->row_lo = 0; d->row_hi = nx + 1; d->own_lo = 1; d->own_hi = nx;
  d->jacobi_iters = 10; d->device = -1; d->flags = 0;
  d->Lx = 0.1; d->Ly = 0.1; d->rho_l = 1000.0; d->rho_g = 50.0; d->nu_l = 1.0e-6; d->nu_g = 1.5e-5;
  d->sigma = 0.007; d->gx = 0; d->gy = -5; d->dt = 4e-6;
  return VOF_OK;
}

int ovof_create(const vof2d_desc* d, void* stream, vof2d_handle* out) {
  (void)stream;
  if (!d || !out || d->abi_version != VOF_ABI_VERSION) return VOF_EINVAL;
  if (d->nx < 3 || d->ny < 3 || d->row_lo < 0 || d->row_hi > d->nx + 1 || d->row_hi - d->row_lo < 2)
    return VOF_EINVAL;
  if (d->dtype != VOF_F64 && d->dtype != VOF_F32) return VOF_EINVAL;
  struct vof2d_ctx* h = (struct vof2d_ctx*)calloc(1, sizeof(*h));
  if (!h) return VOF_ENOMEM;
  h->d = *d;
  vof_consts_compute(d, &h->c);
  int rc = d->dtype == VOF_F64 ? grid_alloc_f64(&h->g64, d) : grid_alloc_f32(&h->g32, d);
  if (rc) { free(h); return rc; }
  *out = h;
  return VOF_OK;
}

int ovof_destroy(vof2d_handle h) {
  if (!h) return VOF_EINVAL;
  if (h->d.dtype == VOF_F64) grid_free_f64(&h->g64); else grid_free_f32(&h->g32);
  free(h);
  return VOF_OK;
}

int ovof_set_init_F(vof2d_handle h, int32_t ic) {
  if (!h || ic < 1 || ic > 3) return VOF_EINVAL;
  DISPATCH(h, set_init_F, ic, h->d.Lx, h->d.Ly);
  return VOF_OK;
}
int ovof_set_BC(vof2d_handle h) { if (!h) return VOF_EINVAL; DISPATCH(h, set_BC); return VOF_OK; }
int ovof_cal_nu_rho(vof2d_handle h) { if (!h) return VOF_EINVAL; DISPATCH(h, cal_nu_rho); return VOF_OK; }
int ovof_get_normal_young(vof2d_handle h) { if (!h) return VOF_EINVAL; DISPATCH(h, get_normal_young); return VOF_OK; }
int ovof_advect_upwind(vof2d_handle h) { if (!h) return VOF_EINVAL; DISPATCH(h, advect_upwind); return VOF_OK; }
int ovof_solve_p_jacobi(vof2d_handle h, int32_t n) {
  if (!h || n < 0) return VOF_EINVAL;
  for (int k = 0; k < n; ++k) DISPATCH(h, solve_p_jacobi, NULL);
  return VOF_OK;
}
double ovof_residual_value(double max_update, double max_p, int32_t criterion) {
  if (!(max_update < HUGE_VAL)) return HUGE_VAL;   /* inf or NaN: diverged */
  if (criterion == VOF_RESID_ABS) return max_update;
  /* a finite update over a tiny (or zero) max|p_new| must not read as "diverged": the quotient is
   * clamped to the largest finite double, so only a non-finite UPDATE ever returns +inf */
  const double q = max_update / (max_p > VOF_RESID_TINY ? max_p : VOF_RESID_TINY);
  return q < HUGE_VAL ? q : DBL_MAX;
}
/* Extension: n sweeps; max|p_new - p| and max|p_new| over owned rows of the last one. */
int ovof_jacobi_sweeps_norms(vof2d_handle h, int32_t n, int32_t build_rhs, double* max_update, double* max_p) {
  if (!h || !max_update || !max_p || n < 1) return VOF_EINVAL;
  /* the extension defines rhs on the CURRENT F (rho = f(F)); the sweeps below recompute the
   * iteration-invariant rhs from the rho array every time, like :239-241 */
  if (build_rhs) ovof_cal_nu_rho(h);
  double norms[2] = {0.0, 0.0};
  for (int k = 0; k < n; ++k) DISPATCH(h, solve_p_jacobi, k == n - 1 ? norms : NULL);
  *max_update = norms[0];
  *max_p = norms[1];
  return VOF_OK;
}
int ovof_jacobi_sweeps_residual(vof2d_handle h, int32_t n, int32_t build_rhs, double* residual) {
  double pmax;
  if (!residual) return VOF_EINVAL;
  return ovof_jacobi_sweeps_norms(h, n, build_rhs, residual, &pmax);
}
/* same driver loop as vof_solve_p (include/vof2d.h) */
int ovof_solve_p(vof2d_handle h, double tol, int32_t max_iters, int32_t check_every, int32_t criterion,
                 int32_t* iters_done, double* residual) {
  if (!h || !iters_done || !residual || max_iters < 1 || check_every < 1) return VOF_EINVAL;
  if (criterion != VOF_RESID_ABS && criterion != VOF_RESID_REL) return VOF_EINVAL;
  int done = 0;
  double r = 0.0;
  while (done < max_iters) {
    int n = check_every < max_iters - done ? check_every : max_iters - done;
    double upd = 0.0, pmax = 0.0;
    ovof_jacobi_sweeps_norms(h, n, done == 0, &upd, &pmax);
    done += n;
    r = ovof_residual_value(upd, pmax, criterion);
    if (r <= tol || !(r < HUGE_VAL)) break;
  }
  *iters_done = done;
  *residual = r;
  return VOF_OK;
}
int ovof_solve_p_residual(vof2d_handle h, double tol, int32_t max_iters, int32_t check_every, int32_t* iters_done,
                          double* residual) {
  return ovof_solve_p(h, tol, max_iters, check_every, VOF_RESID_ABS, iters_done, residual);
}
int ovof_update_uv(vof2d_handle h) { if (!h) return VOF_EINVAL; DISPATCH(h, update_uv); return VOF_OK; }
int ovof_fct_x_sweep(vof2d_handle h) { if (!h) return VOF_EINVAL; DISPATCH(h, fct_x_sweep); return VOF_OK; }
int ovof_fct_y_sweep(vof2d_handle h) { if (!h) return VOF_EINVAL; DISPATCH(h, fct_y_sweep); return VOF_OK; }
/* 2dvof.py:312-318 */
int ovof_solve_VOF_rudman(vof2d_handle h, int64_t istep) {
  if (!h) return VOF_EINVAL;
  if (istep % 2 == 0) { ovof_fct_y_sweep(h); ovof_fct_x_sweep(h); }
  else { ovof_fct_x_sweep(h); ovof_fct_y_sweep(h); }
  return VOF_OK;
}
int ovof_post_process_f(vof2d_handle h) { if (!h) return VOF_EINVAL; DISPATCH(h, post_process_f); return VOF_OK; }

/* 2dvof.py:505-528 */
int ovof_step(vof2d_handle h, int64_t nsteps) {
  if (!h || nsteps < 0) return VOF_EINVAL;
  for (int64_t s = 0; s < nsteps; ++s) {
    h->istep += 1;
    ovof_cal_nu_rho(h);
    ovof_get_normal_young(h);
    ovof_advect_upwind(h);
    ovof_set_BC(h);
    ovof_solve_p_jacobi(h, h->d.jacobi_iters);
    ovof_update_uv(h);
    ovof_set_BC(h);
    ovof_solve_VOF_rudman(h, h->istep);
    ovof_post_process_f(h);
    ovof_set_BC(h);
  }
  return VOF_OK;
}
/* The step cut at the points where p / u,v / F become final, with each field's boundary condition
 * applied once (the schedule of the HIP library's vof_step_phase, include/vof2d.h): phase 0 ends
 * with p (and F) ghosts set, phase 1 = update_uv + u,v ghosts + the first FCT sweep, phase 2 = the
 * second sweep + post_process_f + F ghosts.
 * Test double for the overlapped halo exchange; ovof_step above stays the literal main loop. */
int ovof_step_phase(vof2d_handle h, int32_t phase) {
  if (!h || phase < 0 || phase > 2) return VOF_EINVAL;
  if (phase != h->next_phase) return VOF_ESTATE;
  h->next_phase = phase == 2 ? 0 : phase + 1;
  const int y_first = (h->istep + (phase == 0)) % 2 == 0; /* 2dvof.py:312-318 */
  if (phase == 0) {
    h->istep += 1;
    ovof_cal_nu_rho(h);
    ovof_get_normal_young(h);
    ovof_advect_upwind(h);
    ovof_solve_p_jacobi(h, h->d.jacobi_iters);
    DISPATCH(h, set_BC_mask, 4 | 2);
  } else if (phase == 1) {
    ovof_update_uv(h);
    DISPATCH(h, set_BC_mask, 1 | 8);
    if (y_first) ovof_fct_y_sweep(h); else ovof_fct_x_sweep(h);
  } else {
    if (y_first) ovof_fct_x_sweep(h); else ovof_fct_y_sweep(h);
    /* post_process_f on the interior (on ghosts it is dead: the F boundary condition follows) */
    ovof_post_process_f(h);
    DISPATCH(h, set_BC_mask, 2);
  }
  return VOF_OK;
}
int ovof_get_istep(vof2d_handle h, int64_t* istep) { if (!h || !istep) return VOF_EINVAL; *istep = h->istep; return VOF_OK; }
int ovof_set_istep(vof2d_handle h, int64_t istep) { if (!h) return VOF_EINVAL; h->istep = istep; return VOF_OK; }

static void* field_ptr(vof2d_handle h, const char* name, size_t* esz, int* nr, int* nc) {
  if (h->d.dtype == VOF_F64) {
    *esz = 8; *nr = h->g64.nr; *nc = h->g64.nc;
    return field_f64(&h->g64, name);
  }
  *esz = 4; *nr = h->g32.nr; *nc = h->g32.nc;
  return field_f32(&h->g32, name);
}

int ovof_get_rows(vof2d_handle h, const char* name, int32_t g0, int32_t g1, void* dst, size_t nbytes) {
  size_t esz; int nr, nc;
  if (!h || !name || !dst) return VOF_EINVAL;
  char* f = (char*)field_ptr(h, name, &esz, &nr, &nc);
  if (!f || g0 < h->d.row_lo || g1 > h->d.row_hi || g1 < g0) return VOF_EINVAL;
  size_t n = (size_t)(g1 - g0 + 1) * nc * esz;
  if (n != nbytes) return VOF_EINVAL;
  memcpy(dst, f + (size_t)(g0 - h->d.row_lo) * nc * esz, n);
  return VOF_OK;
}
int ovof_set_rows(vof2d_handle h, const char* name, int32_t g0, int32_t g1, const void* src, size_t nbytes) {
  size_t esz; int nr, nc;
  if (!h || !name || !src) return VOF_EINVAL;
  char* f = (char*)field_ptr(h, name, &esz, &nr, &nc);
  if (!f || g0 < h->d.row_lo || g1 > h->d.row_hi || g1 < g0) return VOF_EINVAL;
  size_t n = (size_t)(g1 - g0 + 1) * nc * esz;
  if (n != nbytes) return VOF_EINVAL;
  memcpy(f + (size_t)(g0 - h->d.row_lo) * nc * esz, src, n);
  return VOF_OK;
}
int ovof_get_field(vof2d_handle h, const char* name, void* dst, size_t nbytes) {
  if (!h) return VOF_EINVAL;
  return ovof_get_rows(h, name, h->d.row_lo, h->d.row_hi, dst, nbytes);
}
int ovof_set_field(vof2d_handle h, const char* name, const void* src, size_t nbytes) {
  if (!h) return VOF_EINVAL;
  return ovof_set_rows(h, name, h->d.row_lo, h->d.row_hi, src, nbytes);
}
int ovof_field_view(vof2d_handle h, const char* name, void** base, int64_t* pitch, int64_t* col0, int64_t* nrows) {
  size_t esz; int nr, nc;
  if (!h || !name) return VOF_EINVAL;
  void* f = field_ptr(h, name, &esz, &nr, &nc);
  if (!f) return VOF_EINVAL;
  if (base) *base = f;
  if (pitch) *pitch = nc;
  if (col0) *col0 = 0;
  if (nrows) *nrows = nr;
  return VOF_OK;
}
int ovof_copy_rows(vof2d_handle dst, vof2d_handle src, const char* name, int32_t g0, int32_t g1) {
  size_t esz, esz2; int nr, nc, nr2, nc2;
  if (!dst || !src || !name) return VOF_EINVAL;
  char* fs = (char*)field_ptr(src, name, &esz, &nr, &nc);
  char* fd = (char*)field_ptr(dst, name, &esz2, &nr2, &nc2);
  if (!fs || !fd || esz != esz2 || nc != nc2) return VOF_EINVAL;
  if (g0 < src->d.row_lo || g1 > src->d.row_hi || g0 < dst->d.row_lo || g1 > dst->d.row_hi || g1 < g0)
    return VOF_EINVAL;
  memcpy(fd + (size_t)(g0 - dst->d.row_lo) * nc * esz, fs + (size_t)(g0 - src->d.row_lo) * nc * esz,
         (size_t)(g1 - g0 + 1) * nc * esz);
  return VOF_OK;
}

int ovof_get_vis_field(vof2d_handle h, const char* which, void* dst, size_t nbytes) {
  if (!h || !which || !dst) return VOF_EINVAL;
  if (h->d.row_lo != 0 || h->d.row_hi != h->d.nx + 1) return VOF_ESTATE;
  int mode = !strcmp(which, "vof") ? 0 : !strcmp(which, "u") ? 1 : !strcmp(which, "v") ? 2 : !strcmp(which, "vnorm") ? 3 : -1;
  if (mode < 0) return VOF_EINVAL;
  size_t esz = h->d.dtype == VOF_F64 ? 8 : 4;
  if (nbytes != (size_t)4 * h->d.nx * h->d.ny * esz) return VOF_EINVAL;
  if (h->d.dtype == VOF_F64) vis_field_f64(&h->g64, mode, h->d.Lx, h->d.Ly, (double*)dst);
  else vis_field_f32(&h->g32, mode, h->d.Lx, h->d.Ly, (float*)dst);
  return VOF_OK;
}
int ovof_interp_velocity(vof2d_handle h, void* dst, size_t nbytes) {
  if (!h || !dst) return VOF_EINVAL;
  if (h->d.row_lo != 0 || h->d.row_hi != h->d.nx + 1) return VOF_ESTATE;
  size_t esz = h->d.dtype == VOF_F64 ? 8 : 4;
  if (nbytes != (size_t)2 * (h->d.nx + 2) * (h->d.ny + 2) * esz) return VOF_EINVAL;
  if (h->d.dtype == VOF_F64) interp_velocity_f64(&h->g64, (double*)dst);
  else interp_velocity_f32(&h->g32, (float*)dst);
  return VOF_OK;
}
int ovof_set_param(vof2d_handle h, const char* name, double value) {
  if (!h || !name) return VOF_EINVAL;
  if (!strcmp(name, "sigma")) {
    h->d.sigma = value; h->c.sigma = value;
    if (h->d.dtype == VOF_F64) h->g64.sigma = value; else h->g32.sigma = (float)value;
    return VOF_OK;
  }
  return VOF_EINVAL;
}
int ovof_get_param(vof2d_handle h, const char* name, double* value) {
  if (!h || !name || !value) return VOF_EINVAL;
#define P(n) if (!strcmp(name, #n)) { *value = h->c.n; return VOF_OK; }
  P(sigma) P(dt) P(dx) P(dy) P(dxi) P(dyi) P(dxi2) P(dyi2) P(rho_l) P(rho_g) P(nu_l) P(nu_g) P(gx) P(gy)
  P(nrm_x) P(nrm_y) P(kap_x) P(kap_y) P(dxdy) P(dtdy) P(dtdx) P(cfl_x) P(cfl_y) P(half_dx) P(half_dy)
  P(sqrt2dx) P(tiny)
#undef P
  if (!strcmp(name, "Lx")) { *value = h->d.Lx; return VOF_OK; }
  if (!strcmp(name, "Ly")) { *value = h->d.Ly; return VOF_OK; }
  return VOF_EINVAL;
}
int ovof_get_counter(vof2d_handle h, const char* name, int64_t* value) {
  if (!h || !name || !value) return VOF_EINVAL;
  if (!strcmp(name, "courant_violations")) {
    *value = h->d.dtype == VOF_F64 ? h->g64.courant : h->g32.courant;
    return VOF_OK;
  }
  return VOF_EINVAL;
}
int ovof_sync(vof2d_handle h) { return h ? VOF_OK : VOF_EINVAL; }
const char* ovof_last_error(vof2d_handle h) { return h ? h->err : "null handle"; }
const char* ovof_backend(void) { return "cpu-oracle"; }

/* node coordinate helper exported for the tests (x[k], y[k] of 2dvof.py:41-46) */
double ovof_node_coord(double L, int32_t n, int32_t k, int32_t cast_f32) {
  return vof_node_coord(L, n, k, cast_f32);
}
