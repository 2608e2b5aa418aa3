// vof2d_kernels.h -- hand-written gfx950 kernels for the 2-D VOF hot path.
//
// Every kernel cites the reference lines (/root/reference/2dvof.py) it
// replaces.  Arithmetic follows the reference's Python expression order,
// left to right, with FMA contraction disabled (-ffp-contract=off), so the
// results equal the CPU oracle's value for value.
//
// All stencil kernels share one decomposition (vof2d_device.h): a wave owns
// 64*V contiguous columns and marches along i with a register window.
//
// The kernels live in kernels/, one file per family:
//   common.h     tile rows, streaming loads / stores, DPP neighbours, exact division
//   verbs.h      one kernel per reference verb (the literal main loop, 2dvof.py:513-528)
//   plan.h       the equal-cost work plan of the fused Jacobi launches
//   momentum.h   k_momentum      (cal_nu_rho + get_normal_young + advect_upwind + rhs)
//   jacobi.h     k_jacobi, k_jacobi_tb   (solve_p_jacobi, the north-star kernel)
//   jacobi_pair.h  k_jacobi_pair   (two k_jacobi_tb launches as one: pairs of waves, result and rhs rows through LDS)
//   transport.h  k_fct_x, k_fct_y, k_transport   (update_uv + solve_VOF_rudman + post_process_f)
//   fused_tm.h   k_tm   (k_transport of one step + k_momentum of the next, rows handed over through LDS)
#pragma once
#include "kernels/common.h"
#include "kernels/verbs.h"
#include "kernels/plan.h"
#include "kernels/momentum.h"
#include "kernels/jacobi.h"
#include "kernels/jacobi_pair.h"
#include "kernels/transport.h"
#include "kernels/fused_tm.h"
