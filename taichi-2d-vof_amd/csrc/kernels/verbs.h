// kernels/verbs.h -- one kernel per reference verb: set_init_F, set_BC, cal_nu_rho, post_process_f, the display fields, get_normal_young, advect_upwind, the rhs of solve_p_jacobi, update_uv
//
// Part of the gfx950 kernel set of the 2-D VOF hot path (see vof2d_kernels.h for the conventions:
// reference line citations, expression order, one wave = 64*V columns marching along i).
#pragma once
#include "common.h"

namespace vof {

// ------------------------------------------------------------------ init
// 2dvof.py:102-134 find_area
template <typename T>
__device__ T find_area(const Consts<T>& c, int i, int j, T cx, T cy, T r) {
  T a;
  T xct = (T)(i - 1) * c.dx + c.half_dx;
  T yct = (T)(j - 1) * c.dy + c.half_dy;
  T xlu = xct - c.half_dx, ylu = yct + c.half_dy;
  T xld = xct - c.half_dx, yld = yct - c.half_dy;
  T xru = xct + c.half_dx, yru = yct + c.half_dy;
  T xrd = xct + c.half_dx, yrd = yct - c.half_dy;
#define VOF_DIST(X, Y) dsqrt<T>(((X) - cx) * ((X) - cx) + ((Y) - cy) * ((Y) - cy))
  T dct = VOF_DIST(xct, yct), dlu = VOF_DIST(xlu, ylu), dld = VOF_DIST(xld, yld), dru = VOF_DIST(xru, yru),
    drd = VOF_DIST(xrd, yrd);
#undef VOF_DIST
  if (dlu > r && dld > r && dru > r && drd > r)
    a = (T)1.0;
  else if (dlu < r && dld < r && dru < r && drd < r)
    a = (T)0.0;
  else {
    a = (T)0.5 + (T)0.5 * (dct - r) / c.sqrt2dx;
    a = var3(a, (T)0, (T)1);
  }
  return a;
}

// node coordinate x[k] of 2dvof.py:43-46: hstack((0, linspace(0, L, n+1), L)).astype(f32)
__device__ __forceinline__ double node_coord(double L, int n, int k, int cast_f32) {
  double v = k == 0 ? 0.0 : (k >= n + 1 ? L : (double)(k - 1) * (L / (double)n));
  if (cast_f32) v = (double)(float)v;
  return v;
}

// 2dvof.py:137-159 set_init_F, all stored cells incl. ghosts; writes F and its sweep twin
template <typename T>
__global__ __launch_bounds__(256) void k_init_F(Geom g, Consts<T> c, T* __restrict__ F, T* __restrict__ F2,
                                                 int ic, double Lx, double Ly, int cast_f32) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = g.row_lo + blockIdx.y;
  if (j > g.ny + 1 || i > g.row_hi) return;
  const size_t o = at(g, i, j);
  T val = F[o];
  if (ic == 1) {
    T xi = (T)node_coord(Lx, g.nx, i, cast_f32), yj = (T)node_coord(Ly, g.ny, j, cast_f32);
    if (xi >= (T)0.0 && xi <= c.ic1_x2 && yj >= (T)0.0 && yj <= c.ic1_y2) val = (T)1.0;
  } else if (ic == 2) {
    val = find_area<T>(c, i, j, c.ic_cx, c.ic2_cy, c.ic_r);
  } else {
    val = (T)1.0 - find_area<T>(c, i, j, c.ic_cx, c.ic3_cy, c.ic_r);
    T yj = (T)node_coord(Ly, g.ny, j, cast_f32);
    if (yj < c.ic3_pool) val = (T)1.0;
  }
  F[o] = val;
  F2[o] = val;
}

// ------------------------------------------------------------------ set_BC
// 2dvof.py:162-189.  One thread per row index (loop 1) and per column index
// (loop 2).  Loop 2 reads are redirected to cells loop 1 does not write, and
// loop 1 skips the cells loop 2 overwrites, so one launch reproduces the
// sequential "loop 1 then loop 2" result (corners take loop-2 values, S11).
// F ghosts are mirrored into the sweep twin F2 (see k_fct_*).
// MASK selects the fields (BC_UV | BC_F | BC_P | BC_RHO): the fused step applies each field's
// boundary condition once, right after the field is final (DESIGN.md "schedule").
enum : int { BC_UV = 1, BC_F = 2, BC_P = 4, BC_RHO = 8, BC_ALL = 7 };
template <typename T, int MASK>
__global__ __launch_bounds__(256) void k_set_bc(Geom g, T* __restrict__ u, T* __restrict__ v, T* __restrict__ F,
                                                 T* __restrict__ F2, T* __restrict__ p, T* __restrict__ rho,
                                                 int r0, int r1) {
  constexpr bool UV = MASK & BC_UV, DF = MASK & BC_F, DP = MASK & BC_P, STORED = MASK & BC_RHO;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int ny = g.ny, nx = g.nx;
  // loop 1: row i, restricted to [r0, r1] (a strip leaves the halo rows of a field whose exchange
  // is in flight to the sender, who ships its rows with their ghost columns)
  const int i = g.row_lo + t;
  if (i <= g.row_hi && i >= r0 && i <= r1) {
    const bool wall_row = (g.wall_lo && i == 1) || (g.wall_hi && i == nx + 1);    // u zeroed by loop 2
    const bool ghost_row = (g.wall_lo && i == 0) || (g.wall_hi && i == nx + 1);   // F,p,v,rho from loop 2
    const size_t a0 = at(g, i, 0), a1 = at(g, i, 1), b0 = at(g, i, ny), b1 = at(g, i, ny + 1);
    if (UV && !wall_row) {
      u[a0] = u[a1];
      u[b1] = u[b0];
    }
    if (!ghost_row) {
      if (UV) {
        v[a1] = (T)0;
        v[b1] = (T)0;
      }
      if (DF) {
        T f0 = F[a1], f1 = F[b0];
        F[a0] = f0; F[b1] = f1;
        F2[a0] = f0; F2[b1] = f1;
      }
      if (DP) {
        p[a0] = p[a1];
        p[b1] = p[b0];
      }
      if (STORED) {
        rho[a0] = rho[a1];
        rho[b1] = rho[b0];
      }
    }
  }
  // loop 2: column j
  const int j = t;
  if (j <= ny + 1) {
    const int jj = j == 0 ? 1 : (j == ny + 1 ? ny : j);  // value loop 1 leaves at column j
    const bool vz = (j == 1 || j == ny + 1);             // loop 1 zeroed v there
    if (g.wall_lo) {
      if (UV) {
        u[at(g, 1, j)] = (T)0;
        v[at(g, 0, j)] = vz ? (T)0 : v[at(g, 1, j)];
      }
      if (DF) {
        T f = F[at(g, 1, jj)];
        F[at(g, 0, j)] = f;
        F2[at(g, 0, j)] = f;
      }
      if (DP) p[at(g, 0, j)] = p[at(g, 1, jj)];
      if (STORED) rho[at(g, 0, j)] = rho[at(g, 1, jj)];
    }
    if (g.wall_hi) {
      if (UV) {
        u[at(g, nx + 1, j)] = (T)0;
        v[at(g, nx + 1, j)] = vz ? (T)0 : v[at(g, nx, j)];
      }
      if (DF) {
        T f = F[at(g, nx, jj)];
        F[at(g, nx + 1, j)] = f;
        F2[at(g, nx + 1, j)] = f;
      }
      if (DP) p[at(g, nx + 1, j)] = p[at(g, nx, jj)];
      if (STORED) rho[at(g, nx + 1, j)] = rho[at(g, nx, jj)];
    }
  }
}


// ghost columns of one F buffer for rows [r0, r1]: the F part of set_BC's loop 1 (:162-174) for the
// edge bands of a strip, whose final F leaves for the neighbour before the rest of the rows exist
template <typename T>
__global__ __launch_bounds__(256) void k_bc_F_cols(Geom g, T* __restrict__ F, int r0, int r1) {
  const int i = r0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (i > r1) return;
  F[at(g, i, 0)] = F[at(g, i, 1)];
  F[at(g, i, g.ny + 1)] = F[at(g, i, g.ny)];
}

// ------------------------------------------------------------------ cal_nu_rho
// 2dvof.py:198-203 (verb only: the fused step recomputes rho/nu from F in place)
template <typename T>
__global__ __launch_bounds__(256) void k_nu_rho(Geom g, Consts<T> c, const T* __restrict__ F, T* __restrict__ rho,
                                                 T* __restrict__ nu) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = g.row_lo + blockIdx.y;
  if (j > g.ny + 1 || i > g.row_hi) return;
  const size_t o = at(g, i, j);
  T f = F[o];
  rho[o] = rho_of(c, f);
  nu[o] = nu_of(c, f);
}

// 2dvof.py:452-455 post_process_f on all stored cells (verb); keeps the twin in sync
template <typename T>
__global__ __launch_bounds__(256) void k_post(Geom g, T* __restrict__ F, T* __restrict__ F2) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = g.row_lo + blockIdx.y;
  if (j > g.ny + 1 || i > g.row_hi) return;
  const size_t o = at(g, i, j);
  T f = var3(F[o], (T)0, (T)1);
  F[o] = f;
  F2[o] = f;
}

// ------------------------------------------------------------------ visualisation fields
// 2dvof.py:458-485 get_vof_field / get_u_field / get_v_field / get_vnorm_field: the (2nx, 2ny)
// image rgb_buf[I] = field[I // r] (r = 2), i.e. the *stored* entries [0, nx) x [0, ny) -- ghost
// index 0 included, nx and nx+1 not -- each repeated 2 x 2.  mode 0: F; 1: u / (Lx/0.2);
// 2: v / (Ly/0.2); 3: sqrt(u^2 + v^2) / (Ly/0.2).  img is dense, row-major (2nx, 2ny).
template <typename T>
__global__ __launch_bounds__(256) void k_vis_field(Geom g, const T* __restrict__ F, const T* __restrict__ u,
                                                    const T* __restrict__ v, T* __restrict__ img, int mode,
                                                    T umax, T vmax) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;  // image column index (y)
  const int a = blockIdx.y;                             // image row index (x)
  if (b >= 2 * g.ny || a >= 2 * g.nx) return;
  const size_t o = at(g, a / 2, b / 2);
  T val;
  if (mode == 0) val = F[o];
  else if (mode == 1) val = u[o] / umax;
  else if (mode == 2) val = v[o] / vmax;
  else val = dsqrt<T>(u[o] * u[o] + v[o] * v[o]) / vmax;
  img[(size_t)a * (size_t)(2 * g.ny) + b] = val;
}

// 2dvof.py:488-492 interp_velocity: V[i,j] = ((u[i,j]+u[i+1,j])/2, (v[i,j]+v[i,j+1])/2) for
// i in [1, nx+1], j in [1, ny].  At i = nx+1 the reference indexes u[nx+2, j], one row past the
// field (undefined in Taichi's release mode); it reads as 0 here.  out is dense (nx+2, ny+2, 2),
// entries outside the loop range stay 0 like the zero-initialised ti.Vector.field.
template <typename T>
__global__ __launch_bounds__(256) void k_interp_velocity(Geom g, const T* __restrict__ u, const T* __restrict__ v,
                                                          T* __restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (j > g.ny + 1 || i > g.nx + 1) return;
  T vx = (T)0, vy = (T)0;
  if (i >= 1 && j >= 1 && j <= g.ny) {
    const T unext = i + 1 <= g.nx + 1 ? u[at(g, i + 1, j)] : (T)0;
    vx = (u[at(g, i, j)] + unext) / (T)2;
    vy = (v[at(g, i, j)] + v[at(g, i, j + 1)]) / (T)2;
  }
  const size_t o = ((size_t)i * (size_t)(g.ny + 2) + j) * 2;
  out[o] = vx;
  out[o + 1] = vy;
}

// ------------------------------------------------------------------ normals
// 2dvof.py:285-306 get_normal_young loop 1: F (3x3) -> mx, my on interior rows.
template <typename T, int V>
__global__ __launch_bounds__(256) void k_normals(Geom g, Consts<T> c, const T* __restrict__ F, T* __restrict__ mx,
                                                  T* __restrict__ my, int R) {
  int j0, ra, rb;
  if (!wave_tile<V>(g, g.ilo, g.ihi, R, j0, ra, rb)) return;
  const T cxn = c.nrm_x, cyn = c.nrm_y;
  size_t o = at(g, ra, j0);
  Row<T, V> m, z, p;  // rows i-1, i, i+1
  load_row<T, V>(m, F + o - g.pitch);
  load_row<T, V>(z, F + o);
  for (int i = ra; i <= rb; ++i, o += g.pitch) {
    load_row<T, V>(p, F + o + g.pitch);
    T ox[V], oy[V];
#pragma unroll
    for (int q = 0; q < V; ++q) {
      const T Fmm = left_of(m, q), Fm0 = m.c[q], Fmp = right_of(m, q);
      const T F0m = left_of(z, q), F00 = z.c[q], F0p = right_of(z, q);
      const T Fpm = left_of(p, q), Fp0 = p.c[q], Fpp = right_of(p, q);
      T mx1 = cxn * (Fpp + Fp0 - F0p - F00);
      T my1 = cyn * (Fpp - Fp0 + F0p - F00);
      T mx2 = cxn * (Fp0 + Fpm - F00 - F0m);
      T my2 = cyn * (Fp0 - Fpm + F00 - F0m);
      T mx3 = cxn * (F00 + F0m - Fm0 - Fmm);
      T my3 = cyn * (F00 - F0m + Fm0 - Fmm);
      T mx4 = cxn * (F0p + F00 - Fmp - Fm0);
      T my4 = cyn * (F0p - F00 + Fmp - Fm0);
      T mxsum = (mx1 + mx2 + mx3 + mx4) / (T)4;
      T mysum = (my1 + my2 + my3 + my4) / (T)4;
      if (dabs<T>(mxsum) < c.tiny && dabs<T>(mysum) < c.tiny) {
        ox[q] = mxsum;
        oy[q] = mysum;
      } else {
        T magnitude = dsqrt<T>(mxsum * mxsum + mysum * mysum);
        ox[q] = mxsum / magnitude;
        oy[q] = mysum / magnitude;
      }
    }
    store_c<T, V>(mx + o, ox, j0, 1, g.ny);
    store_c<T, V>(my + o, oy, j0, 1, g.ny);
    m = z;
    z = p;
  }
}

// 2dvof.py:307-309 get_normal_young loop 2: kappa from mx (i+-1) and my (j+-1)
template <typename T, int V>
__global__ __launch_bounds__(256) void k_kappa(Geom g, Consts<T> c, const T* __restrict__ mx,
                                                const T* __restrict__ my, T* __restrict__ kappa, int R) {
  int j0, ra, rb;
  if (!wave_tile<V>(g, g.ilo, g.ihi, R, j0, ra, rb)) return;
  size_t o = at(g, ra, j0);
  T xm[V], xz[V], xp[V];
  load_c<T, V>(xm, mx + o - g.pitch);
  load_c<T, V>(xz, mx + o);
  for (int i = ra; i <= rb; ++i, o += g.pitch) {
    load_c<T, V>(xp, mx + o + g.pitch);
    Row<T, V> y;
    load_row<T, V>(y, my + o);
    T k[V];
#pragma unroll
    for (int q = 0; q < V; ++q)
      k[q] = -(c.kap_x * (xp[q] - xm[q]) + c.kap_y * (right_of(y, q) - left_of(y, q)));
    store_c<T, V>(kappa + o, k, j0, 1, g.ny);
#pragma unroll
    for (int q = 0; q < V; ++q) {
      xm[q] = xz[q];
      xz[q] = xp[q];
    }
  }
}

// ------------------------------------------------------------------ predictor
// 2dvof.py:206-233 advect_upwind: u*, v* from u, v, kappa, F (rho, nu).
// STORED: read the rho / nu arrays written by cal_nu_rho (verb semantics);
// otherwise recompute them from F per cell (identical values: rho[i,j] is a
// pure function of F[i,j] and F is unchanged since cal_nu_rho, 2dvof.py:513-517).
template <typename T, int V, bool STORED>
__global__ __launch_bounds__(256) void k_predictor(Geom g, Consts<T> c, const T* __restrict__ u,
                                                    const T* __restrict__ v, const T* __restrict__ kappa,
                                                    const T* __restrict__ F, const T* __restrict__ rho,
                                                    const T* __restrict__ nu, T* __restrict__ us,
                                                    T* __restrict__ vs, int R) {
  int j0, ra, rb;
  if (!wave_tile<V>(g, g.ilo, g.ihi, R, j0, ra, rb)) return;
  const T dt = c.dt, dxi = c.dxi, dyi = c.dyi, dxi2 = c.dxi2, dyi2 = c.dyi2;
  size_t o = at(g, ra, j0);
  Row<T, V> um, uz, up, vm, vz, vp;
  T km[V], Fm[V], rm_[V];
  load_row<T, V>(um, u + o - g.pitch);
  load_row<T, V>(uz, u + o);
  load_row<T, V>(vm, v + o - g.pitch);
  load_row<T, V>(vz, v + o);
  load_c<T, V>(km, kappa + o - g.pitch);
  load_c<T, V>(Fm, F + o - g.pitch);
  if (STORED) load_c<T, V>(rm_, rho + o - g.pitch);
  for (int i = ra; i <= rb; ++i, o += g.pitch) {
    load_row<T, V>(up, u + o + g.pitch);
    load_row<T, V>(vp, v + o + g.pitch);
    Row<T, V> kz, Fz, rz;
    T nz[V];
    load_c<T, V>(kz.c, kappa + o);
    kz.l = kappa[o - 1];
    load_c<T, V>(Fz.c, F + o);
    Fz.l = F[o - 1];
    if (STORED) {
      load_c<T, V>(rz.c, rho + o);
      rz.l = rho[o - 1];
      load_c<T, V>(nz, nu + o);
    }
    T ou[V], ov[V];
#pragma unroll
    for (int q = 0; q < V; ++q) {
      const T u00 = uz.c[q], um0 = um.c[q], up0 = up.c[q], u0m = left_of(uz, q), u0p = right_of(uz, q);
      const T upm = left_of(up, q);
      const T v00 = vz.c[q], vm0 = vm.c[q], vp0 = vp.c[q], v0m = left_of(vz, q), v0p = right_of(vz, q);
      const T vmp = right_of(vm, q);
      const T F00 = Fz.c[q], Fm0 = Fm[q], F0m = left_of(Fz, q);
      const T k00 = kz.c[q], km0 = km[q], k0m = left_of(kz, q);
      T rho00, rhom0, rho0m, nu00;
      if (STORED) {
        rho00 = rz.c[q]; rhom0 = rm_[q]; rho0m = left_of(rz, q); nu00 = nz[q];
      } else {
        rho00 = rho_of(c, F00); rhom0 = rho_of(c, Fm0); rho0m = rho_of(c, F0m); nu00 = nu_of(c, F00);
      }
      {  // :208-220
        T v_here = (T)0.25 * (vm0 + vmp + v00 + v0p);
        T dudx = u00 > 0 ? (u00 - um0) * dxi : (up0 - u00) * dxi;
        T dudy = v_here > 0 ? (u00 - u0m) * dyi : (u0p - u00) * dyi;
        T kappa_ave = (k00 + km0) / (T)2.0;
        T fx_kappa = -c.sigma * (F00 - Fm0) * kappa_ave / c.dx;
        ou[q] = (u00 + dt * (nu00 * (um0 - (T)2 * u00 + up0) * dxi2 + nu00 * (u0m - (T)2 * u00 + u0p) * dyi2 -
                             u00 * dudx - v_here * dudy + c.gx + fx_kappa * (T)2 / (rho00 + rhom0)));
      }
      {  // :221-233
        T u_here = (T)0.25 * (u0m + u00 + upm + up0);
        T dvdx = u_here > 0 ? (v00 - vm0) * dxi : (vp0 - v00) * dxi;
        T dvdy = v00 > 0 ? (v00 - v0m) * dyi : (v0p - v00) * dyi;
        T kappa_ave = (k00 + k0m) / (T)2.0;
        T fy_kappa = -c.sigma * (F00 - F0m) * kappa_ave / c.dy;
        ov[q] = (v00 + dt * (nu00 * (vm0 - (T)2 * v00 + vp0) * dxi2 + nu00 * (v0m - (T)2 * v00 + v0p) * dyi2 -
                             u_here * dvdx - v00 * dvdy + c.gy + fy_kappa * (T)2 / (rho00 + rho0m)));
      }
    }
    if (i >= 2) store_c<T, V>(us + o, ou, j0, 1, g.ny);  // i in [imin+1, imax]
    store_c<T, V>(vs + o, ov, j0, 2, g.ny);               // j in [jmin+1, jmax]
    um = uz; uz = up; vm = vz; vz = vp;
#pragma unroll
    for (int q = 0; q < V; ++q) {
      km[q] = kz.c[q];
      Fm[q] = Fz.c[q];
      if (STORED) rm_[q] = rz.c[q];
    }
  }
}


// ------------------------------------------------------------------ rhs
// 2dvof.py:239-241, hoisted out of the Jacobi loop (it does not depend on p;
// precedent: cal_velocity_div, diff_vof_replaced.py:277-282).
template <typename T, int V, bool STORED>
__global__ __launch_bounds__(256) void k_rhs(Geom g, Consts<T> c, const T* __restrict__ us,
                                              const T* __restrict__ vs, const T* __restrict__ F,
                                              const T* __restrict__ rho, T* __restrict__ rhs, int R) {
  int j0, ra, rb;
  if (!wave_tile<V>(g, g.ilo, g.ihi, R, j0, ra, rb)) return;
  size_t o = at(g, ra, j0);
  T uz[V], up[V];
  load_c<T, V>(uz, us + o);
  for (int i = ra; i <= rb; ++i, o += g.pitch) {
    load_c<T, V>(up, us + o + g.pitch);
    Row<T, V> vz;
    load_c<T, V>(vz.c, vs + o);
    vz.r = vs[o + V];
    T f[V], out[V];
    load_c<T, V>(f, STORED ? rho + o : F + o);
#pragma unroll
    for (int q = 0; q < V; ++q) {
      T r = STORED ? f[q] : rho_of(c, f[q]);
      out[q] = r / c.dt * ((up[q] - uz[q]) * c.dxi + (right_of(vz, q) - vz.c[q]) * c.dyi);
    }
    store_c<T, V>(rhs + o, out, j0, 1, g.ny);
#pragma unroll
    for (int q = 0; q < V; ++q) uz[q] = up[q];
  }
}


// ------------------------------------------------------------------ corrector
// 2dvof.py:269-280 update_uv (+ Courant prints -> counter over owned rows)
template <typename T, int V, bool STORED>
__global__ __launch_bounds__(256) void k_correct(Geom g, Consts<T> c, const T* __restrict__ p,
                                                  const T* __restrict__ F, const T* __restrict__ rho,
                                                  const T* __restrict__ us, const T* __restrict__ vs,
                                                  T* __restrict__ u, T* __restrict__ v, int R,
                                                  unsigned long long* __restrict__ courant) {
  int j0, ra, rb;
  if (!wave_tile<V>(g, g.ilo, g.ihi, R, j0, ra, rb)) return;
  size_t o = at(g, ra, j0);
  T pm[V], rm_[V];
  load_c<T, V>(pm, p + o - g.pitch);
  {
    T f[V];
    load_c<T, V>(f, STORED ? rho + o - g.pitch : F + o - g.pitch);
#pragma unroll
    for (int q = 0; q < V; ++q) rm_[q] = STORED ? f[q] : rho_of(c, f[q]);
  }
  unsigned int viol = 0;
  for (int i = ra; i <= rb; ++i, o += g.pitch) {
    Row<T, V> pz, rz;
    load_c<T, V>(pz.c, p + o);
    pz.l = p[o - 1];
    {
      T f[V];
      load_c<T, V>(f, STORED ? rho + o : F + o);
      T fl = STORED ? rho[o - 1] : F[o - 1];
#pragma unroll
      for (int q = 0; q < V; ++q) rz.c[q] = STORED ? f[q] : rho_of(c, f[q]);
      rz.l = STORED ? fl : rho_of(c, fl);
    }
    T usz[V], vsz[V], ou[V], ov[V];
    load_c<T, V>(usz, us + o);
    load_c<T, V>(vsz, vs + o);
    const bool own = i >= g.own_lo && i <= g.own_hi;
#pragma unroll
    for (int q = 0; q < V; ++q) {
      T r = (rz.c[q] + rm_[q]) * (T)0.5;
      ou[q] = usz[q] - c.dt / r * (pz.c[q] - pm[q]) * c.dxi;
      T r2 = (rz.c[q] + left_of(rz, q)) * (T)0.5;
      ov[q] = vsz[q] - c.dt / r2 * (pz.c[q] - left_of(pz, q)) * c.dyi;
      const int j = j0 + q;
      if (own && j <= g.ny) {
        if (i >= 2 && ou[q] * c.dt > c.cfl_x) viol++;
        if (j >= 2 && ov[q] * c.dt > c.cfl_y) viol++;
      }
    }
    if (i >= 2) store_c<T, V>(u + o, ou, j0, 1, g.ny);
    store_c<T, V>(v + o, ov, j0, 2, g.ny);
#pragma unroll
    for (int q = 0; q < V; ++q) {
      pm[q] = pz.c[q];
      rm_[q] = rz.c[q];
    }
  }
  if (__any(viol != 0)) {
    unsigned int tot = viol;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) tot += __shfl_down(tot, s, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(courant, (unsigned long long)tot);
  }
}

// cells of the computable rows whose F is an exact zero (the batch-form rule of vof_step: one look at the state per handle)
template <typename T>
__global__ __launch_bounds__(256) void k_gas_cells(Geom g, const T* __restrict__ F, unsigned long long* __restrict__ out) {
  unsigned int n = 0;
  for (int i = g.ilo + (int)blockIdx.x; i <= g.ihi; i += (int)gridDim.x)
    for (int j = 1 + (int)threadIdx.x; j <= g.ny; j += 256) n += F[at(g, i, j)] == (T)0 ? 1u : 0u;
#pragma unroll
  for (int sft = 32; sft > 0; sft >>= 1) n += __shfl_down(n, sft, 64);
  if ((threadIdx.x & 63) == 0 && n) atomicAdd(out, (unsigned long long)n);
}

}  // namespace vof
