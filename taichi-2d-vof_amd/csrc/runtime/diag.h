// runtime/diag.h -- diagnostic build only (-DVOF_WAVE_TIMES): launches of the pair kernels in their ablated forms
//
// Included by vof2d_api.hip behind selftest.h; never part of the product library.  See vof_debug_time_kernel.
#pragma once
#include "schedule.h"

namespace {

template <int ABL>
void dbg_pair(vof2d_ctx* h, int plan) {
  typedef double T; constexpr int V = VecWidth<T>::V;
  h->jpair_active = true;
  int ntt = 0;
  const int R = L<T>::jacobi_pair_geom(h, ntt);
  const TbPlan tp = L<T>::tb_plan(h, plan ? (int)(h->istep & 1) : -1);
  h->jpair_active = false;
  const unsigned pairs = tp.masks ? (unsigned)tp.waves : (unsigned)(((h->g.ihi - h->g.ilo + R) / R) * ntt);
  launch_block(h, kJacobiPair, k_jacobi_pair<T, V, 5, true, ABL>, dim3(pairs), 128u, 0, h->g, L<T>::C(h), (const T*)F_<T>(h, fP),
               (const T*)F_<T>(h, fRHS), F_<T>(h, fPT), R, ntt, tp, h->g.ilo, h->g.ihi);
}
template <bool YFIRST, int ABL>
void dbg_tm(vof2d_ctx* h) {
  typedef double T; constexpr int V = VecWidth<T>::V;
  constexpr int ST = 64 * V - 2 * TmGeom::HF;
  const int ntf = (h->g.ny + ST - 1) / ST, first = h->g.ilo, last = h->g.ihi;
  const int R = L<T>::tm_chunk_rows(h, last - first + 1, ntf, resident_blocks(h, k_tm<T, V, YFIRST, false, true, ABL>, 128));
  const TbPlan tp{nullptr, nullptr, 0, 0, 0, 0, 0};
  const unsigned pairs = (unsigned)(((last - first + R) / R) * ntf);
  launch_block(h, kTM, k_tm<T, V, YFIRST, false, true, ABL>, dim3(pairs), 128u, 0, h->g, L<T>::C(h), (const T*)F_<T>(h, fF), F_<T>(h, fF2), ntf,
               (const T*)F_<T>(h, fUS), (const T*)F_<T>(h, fVS), (const T*)F_<T>(h, fP), F_<T>(h, fU), F_<T>(h, fV),
               F_<T>(h, fMX), F_<T>(h, fMY), F_<T>(h, fRHS), h->d_courant + 3, R, tp, first, last, 1, 0);
}

}  // namespace
