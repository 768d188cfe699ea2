// awfl_vertical.h -- host-side construction of the per-level vertical WENO matrices (init time only).
//
// Restates dynamics/awfl/Dycore.h:904-937 (normalised cell-edge locations per level, including the reference's
// off-centre indexing: level k is built from cells k-1..k+3) and
// dynamics/awfl/TransformMatrices_variable.h:11-69 (Vandermonde of cell-average monomials, inverted).
// The reference inverts with yakl::intrinsics::matinv_ge, a third-party routine absent from the tree (YAKL
// submodule, version unpinned); here it is Gauss-Jordan elimination without pivoting, (col,row) order.
//
// Output per level (and per ensemble member): VZ_STRIDE = 31 doubles, the difference-form table `DTable` of
// awfl_device.h (make_dtable): lower-candidate x / x^2 / even-edge coefficients and the bridged upper polynomial
// (WenoLimiter.h:128-136 folded in; linear in the stencil, so exact up to rounding).
// The stencil-form vert_sten_to_coefs / vert_weno_recon_lower are also returned for the DataManager entries of
// those names.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#include "awfl_constants.h"
#include "supercell_sounding.h"
#include "awfl_device.h"

namespace pama {

inline void matinv_ge_host(int n, const double *a, double *inv) {
  double scratch[25];
  for (int icol = 0; icol < n; icol++)
    for (int irow = 0; irow < n; irow++) {
      scratch[icol * n + irow] = a[icol * n + irow];
      inv[icol * n + irow] = (icol == irow) ? 1.0 : 0.0;
    }
  for (int d = 0; d < n; d++) {
    double factor = 1.0 / scratch[d * n + d];
    for (int icol = d; icol < n; icol++) scratch[icol * n + d] *= factor;
    for (int icol = 0; icol < n; icol++) inv[icol * n + d] *= factor;
    for (int irow = d + 1; irow < n; irow++) {
      double f = scratch[d * n + irow];
      for (int icol = d; icol < n; icol++) scratch[icol * n + irow] -= f * scratch[icol * n + d];
      for (int icol = 0; icol < n; icol++) inv[icol * n + irow] -= f * inv[icol * n + d];
    }
  }
  for (int d = n - 1; d >= 1; d--)
    for (int irow = 0; irow < d; irow++) {
      double f = scratch[d * n + irow];
      for (int icol = irow + 1; icol < n; icol++) scratch[icol * n + irow] -= f * scratch[icol * n + d];
      for (int icol = 0; icol < n; icol++) inv[icol * n + irow] -= f * inv[icol * n + d];
    }
}

// TransformMatrices_variable.h:11-46
inline void sten_to_coefs_variable_host(int n, const double *locs, double *rslt) {
  double c2s[25], pwr[6];
  for (int i = 0; i < n + 1; i++) pwr[i] = locs[i];
  for (int i = 0; i < n; i++) c2s[i] = 1;
  for (int i = 1; i < n; i++) {
    for (int j = 0; j < n + 1; j++) pwr[j] *= locs[j];
    for (int j = 0; j < n; j++) c2s[i * n + j] = 1. / (i + 1.) * (pwr[j] - pwr[j + 1]) / (locs[j] - locs[j + 1]);
  }
  matinv_ge_host(n, c2s, rslt);
}

// Tables of pow_pos_fast (awfl_device.h), in 80-bit arithmetic: 1/c_i with 24 bits and -log2 of that stored value in two words (the
// high one a multiple of 2^-32), 2^(j/64) in two words.
inline void build_pow_tab(PowTab &T) {
  for (int i = 0; i < 128; i++) {
    const long double c = 1.0L + (i + 0.5L) / 128.0L;
    const double ic = (double)(float)(1.0L / c);
    const long double l2c = -log2l((long double)ic);
    const double lh = (double)(floorl(l2c * 4294967296.0L + 0.5L) / 4294967296.0L);
    T.lg[i].ic = ic; T.lg[i].lh = lh; T.lg[i].ll = (double)(l2c - (long double)lh);
  }
  for (int j = 0; j < 64; j++) {
    const long double v = exp2l(j / 64.0L);
    const double th = (double)v;
    T.ex[j].th = th; T.ex[j].tl = (double)(v - (long double)th);
  }
}

struct VerticalTables {
  bool per_ens;                 // false: all ensemble members share one dz column
  std::vector<double> table;    // (nz+2,VZ_STRIDE) or (nz+2,VZ_STRIDE,nens)
  std::vector<double> s2c;      // vert_sten_to_coefs    (nz+2,5,5,nens)
  std::vector<double> wrl;      // vert_weno_recon_lower (nz+2,3,3,3,nens)
};

inline void level_matrices(const double *dzcol /* stride nens */, long long stride, int nz, int k, double s2c[25],
                           double wrl[27], double dform[VZ_STRIDE]) {
  double dzloc[5], locs[6];
  for (int kk = 0; kk < 5; kk++) {
    int ind1 = std::min(nz - 1, std::max(0, -1 + k + kk));
    int ind2 = std::min(nz - 1, std::max(0, -1 + k));
    dzloc[kk] = dzcol[ind1 * stride] / dzcol[ind2 * stride];
  }
  locs[0] = 0;
  for (int kk = 1; kk < 6; kk++) locs[kk] = locs[kk - 1] + dzloc[kk - 1];
  double midloc = (locs[2] + locs[3]) / 2;
  for (int kk = 0; kk < 6; kk++) locs[kk] = locs[kk] - midloc;
  sten_to_coefs_variable_host(5, locs, s2c);
  for (int i = 0; i < 3; i++) {
    double lo[9];
    sten_to_coefs_variable_host(3, locs + i, lo);
    for (int jj = 0; jj < 3; jj++)
      for (int ii = 0; ii < 3; ii++) wrl[(i * 3 + jj) * 3 + ii] = lo[jj * 3 + ii];
  }
  const double raw[4] = AWFL_WENO_IDL_INIT;
  double sum = ((raw[0] + raw[1]) + raw[2]) + raw[3], idl[4];
  for (int i = 0; i < 4; i++) idl[i] = raw[i] / (sum + 1.0e-20);
  double lo[3][3][3], hi[5][5];
  for (int i = 0; i < 3; i++)
    for (int s = 0; s < 3; s++)
      for (int ii = 0; ii < 3; ii++) lo[i][s][ii] = wrl[(i * 3 + s) * 3 + ii];
  for (int s = 0; s < 5; s++)
    for (int ii = 0; ii < 5; ii++) hi[s][ii] = s2c[s * 5 + ii];
  DTable t = make_dtable(lo, hi, idl, locs[3] - locs[2]);   // width of the centre cell in the matrices' coordinate
  static_assert(sizeof(DTable) == (VZ_STRIDE - 1) * sizeof(double), "DTable layout");
  // stored form: the x^2 rows of the lower candidates times sqrt(13/3) (their TV is then a1^2 + a2'^2), and the matching
  // even-part factor behind the struct (awfl_device.h: DTable, weno5_blend)
  const double sq = std::sqrt(AWFL_TV3_A2A2);
  for (int i = 0; i < 3; i++)
    for (int q = 0; q < 2; q++) t.lo2[i][q] *= sq;
  std::memcpy(dform, &t, sizeof(t));
  dform[VZ_STRIDE - 1] = t.k2 / sq;
}

// dz: host copy of vertical_cell_dz (nz,nens)
inline VerticalTables build_vertical_tables(const double *dz, int nz, int nens) {
  VerticalTables vt;
  vt.per_ens = false;
  for (int k = 0; k < nz && !vt.per_ens; k++)
    for (int e = 1; e < nens; e++)
      if (dz[(long long)k * nens + e] != dz[(long long)k * nens]) { vt.per_ens = true; break; }
  const int nl = nz + 2;
  vt.s2c.assign((size_t)nl * 25 * nens, 0.0);
  vt.wrl.assign((size_t)nl * 27 * nens, 0.0);
  vt.table.assign((size_t)nl * VZ_STRIDE * (vt.per_ens ? nens : 1), 0.0);
  for (int k = 0; k < nl; k++) {
    double s2c[25], wrl[27], br[VZ_STRIDE];
    for (int e = 0; e < nens; e++) {
      if (e == 0 || vt.per_ens) level_matrices(dz + e, nens, nz, k, s2c, wrl, br);
      for (int m = 0; m < 25; m++) vt.s2c[((size_t)k * 25 + m) * nens + e] = s2c[m];
      for (int m = 0; m < 27; m++) vt.wrl[((size_t)k * 27 + m) * nens + e] = wrl[m];
      if (vt.per_ens) {
        for (int m = 0; m < VZ_STRIDE; m++) vt.table[((size_t)k * VZ_STRIDE + m) * nens + e] = br[m];
      } else if (e == 0) {
        for (int m = 0; m < VZ_STRIDE; m++) vt.table[(size_t)k * VZ_STRIDE + m] = br[m];
      }
    }
  }
  return vt;
}

// ---- init_supercell column integration (Dycore.h:1096-1230), host side, once at init; the sounding: supercell_sounding.h
struct SupercellColumns { std::vector<double> hy_dens, hy_pres, dens_vap_gll; };   // (nz,nens), (nz,nens), (nz,9,nens)

// dz, zmid (nz,nens), zint (nz+1,nens): host copies.  Quirk Q10: the reference's cell-mean kernel broadcasts every
// (k, member) result to ALL members through a loop variable that shadows the member index (Dycore.h:1226-1229); in the
// reference's serial order the last member's value wins for every member, which is what is restated here.
inline SupercellColumns supercell_columns(const double *dz, const double *zmid, const double *zint, int nz, int nens,
                                          double R_d, double R_v, double grav, double gamma, double C0) {
  const double p_0 = 100000;
  const double pts[9] = AWFL_GLL9_PTS_INIT, wts[9] = AWFL_GLL9_WTS_INIT;
  SupercellColumns out;
  out.hy_dens.assign((size_t)nz * nens, 0.0);
  out.hy_pres.assign((size_t)nz * nens, 0.0);
  out.dens_vap_gll.assign((size_t)nz * 9 * nens, 0.0);
  std::vector<double> pg((size_t)nz * 9), dg((size_t)nz * 9);
  for (int e = 0; e < nens; e++) {
    const Sounding snd = Sounding::make(zint[(size_t)nz * nens + e], R_d, grav);
    pg[0] = p_0;
    for (int k = 0; k < nz; k++) {
      const double dzk = dz[(size_t)k * nens + e], cellmid = zmid[(size_t)k * nens + e];
      for (int kk = 0; kk < 8; kk++) {
        double b = cellmid + pts[kk] * dzk, t = cellmid + pts[kk + 1] * dzk;
        double m = 0.5 * (b + t), gdz = dzk * (pts[kk + 1] - pts[kk]);
        double tot = 0;
        for (int kkk = 0; kkk < 9; kkk++) {
          double temp, qv = snd.vapour_mixing_ratio(m + gdz * pts[kkk], temp);
          tot += (-(1 + qv) * grav / (R_d + qv * R_v) / temp) * wts[kkk];
        }
        tot *= dzk * (pts[kk + 1] - pts[kk]);
        pg[(size_t)k * 9 + kk + 1] = pg[(size_t)k * 9 + kk] * std::exp(tot);
        if (kk == 7 && k < nz - 1) pg[(size_t)(k + 1) * 9] = pg[(size_t)k * 9 + 8];
      }
    }
    for (int k = 0; k < nz; k++) {
      const double dzk = dz[(size_t)k * nens + e], cellmid = zmid[(size_t)k * nens + e];
      double press_tot = 0, dens_tot = 0;
      for (int kk = 0; kk < 9; kk++) {
        double temp, qv = snd.vapour_mixing_ratio(cellmid + pts[kk] * dzk, temp);
        double press = pg[(size_t)k * 9 + kk];
        double dens_dry = press / (R_d + qv * R_v) / temp, dens_vap = qv * dens_dry;
        out.dens_vap_gll[((size_t)k * 9 + kk) * nens + e] = dens_vap;
        dg[(size_t)k * 9 + kk] = dens_dry + dens_vap;
      }
      for (int kk = 0; kk < 9; kk++) { press_tot += pg[(size_t)k * 9 + kk] * wts[kk]; dens_tot += dg[(size_t)k * 9 + kk] * wts[kk]; }
      for (int e2 = 0; e2 < nens; e2++) { out.hy_dens[(size_t)k * nens + e2] = dens_tot; out.hy_pres[(size_t)k * nens + e2] = press_tot; }
    }
  }
  (void)gamma; (void)C0;
  return out;
}

}  // namespace pama
