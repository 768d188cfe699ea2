/*
 * awfl_oracle.c -- CPU restatement of the reference AWFL dycore step.   TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the MI355X-native AWFL path.  It is NOT part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (pam_amd/csrc) never calls into it and has no CPU fallback.
 *
 * It restates, operation by operation and in the same floating-point association order, what
 * /root/reference/dynamics/awfl/{Dycore.h,WenoLimiter.h,TransformMatrices*.h} compute
 * (E3SM-Project/PAM @ 2025-03-03).  Each function cites the reference lines it follows.
 * Build with -ffp-contract=off so that no FMA contraction changes the rounding.
 *
 * PINNING STATUS: "parity unpinned" by reference-owned fixtures -- the reference holds no golden
 * vector or numeric assertion for this path and cannot be built here (YAKL, its
 * array/launch library, is an absent un-vendored submodule; stand-in builds are not made).  The oracle
 * is checked instead against the reference-arithmetic probe values recorded in SURVEY.md Appendix B
 * (tests/test_oracle_kat.py) and against analytic properties (hydrostatic balance, conservation,
 * WENO order, uniform-grid identity of the variable matrices).
 *
 * Deliberate deviations from the reference (documented in DESIGN.md):
 *   D1 vertical-ghost kernel: the reference reads the potential-temperature ghost of
 *      level hs-1-kk while another iteration writes it (Dycore.h:665-694).  We implement the intended,
 *      order-independent semantics: ghost theta = theta of the boundary cell.
 *   D2 the throw-away banded solve + print in init() (Dycore.h:851-864) is dropped.
 *   D3 yakl::intrinsics::matinv_ge (third party, version unpinned) is restated as Gauss-Jordan
 *      elimination without pivoting, (col,row) index order.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "awfl_constants.h"

#define ORD 5
#define HS 3
#define NUM_STATE 5
#define ID_R 0
#define ID_U 1
#define ID_V 2
#define ID_W 3
#define ID_T 4
#define MAX_TRACERS 50 /* pam_const.h:24 max_fields */

typedef struct awfl_oracle {
  int nens, nx, ny, nz, nt;
  double xlen, ylen;
  /* options (Dycore.h:871-891) */
  double R_d, R_v, cp_d, cp_v, p0, grav, cv_d, gamma_d, kappa_d, cv_v, C0;
  int grav_balance; /* option balance_hydrostasis_with_gravity (Dycore.h:866) */
  int idWV;
  unsigned char tracer_positive[MAX_TRACERS], tracer_adds_mass[MAX_TRACERS];
  double *dz;            /* vertical_cell_dz (nz,nens) */
  double *vert_s2c;      /* vert_sten_to_coefs (nz+2,5,5,nens)      Dycore.h:898 */
  double *vert_wrl;      /* vert_weno_recon_lower (nz+2,3,3,3,nens) Dycore.h:897 */
  double *grav_var;      /* variable_gravity (nz,nens)              Dycore.h:868 */
  double *hy_dens_cells; /* (nz,nens) Dycore.h:983 */
  double *hy_pres_cells; /* (nz,nens) Dycore.h:984 */
  /* constant matrices */
  double s2c[5][5], wrl[3][3][3], c2g[5][2], idl[4], sigma;
  /* debug taps: if non-NULL, compute_tendencies copies fluxes here (post-FCT) */
  double *tap_flux_x, *tap_flux_y, *tap_flux_z;
} awfl_oracle_t;

/* ---------------------------------------------------------------------------------------------- */
/* WenoLimiter.h:22-29 convexify<5>: 4 weights                                                    */
static void convexify4(double w[4]) {
  double sum = 0.0;
  const double eps = 1.0e-20;
  for (int i = 0; i < 4; i++) sum += w[i];
  for (int i = 0; i < 4; i++) w[i] /= (sum + eps);
}

/* WenoLimiter.h:11-19 map_weights<5> */
static void map_weights4(const double idl[4], double w[4]) {
  for (int i = 0; i < 4; i++) {
    w[i] = w[i] * (idl[i] + idl[i] * idl[i] - 3.0 * idl[i] * w[i] + w[i] * w[i]) /
           (idl[i] * idl[i] + w[i] * (1.0 - 2.0 * idl[i]));
  }
}

/* WenoLimiter.h:32-95 wenoSetIdealSigma<5> */
void awfl_oracle_ideal_sigma(double idl[4], double *sigma) {
  const double init[4] = AWFL_WENO_IDL_INIT;
  *sigma = AWFL_WENO_SIGMA;
  for (int i = 0; i < 4; i++) idl[i] = init[i];
  convexify4(idl);
}

/* TransformMatrices.h:188-193 coefs_to_tv<3>; :871-876 coefs_to_tv<5> */
static double tv3(const double a[3]) { return 1.0 * (a[1] * a[1]) + AWFL_TV3_A2A2 * (a[2] * a[2]); }
static double tv5(const double a[5]) {
  return AWFL_TV5_A1A1 * (a[1] * a[1]) + AWFL_TV5_A2A2 * (a[2] * a[2]) + AWFL_TV5_A1A3 * a[1] * a[3] +
         AWFL_TV5_A3A3 * (a[3] * a[3]) + AWFL_TV5_A2A4 * a[2] * a[4] + AWFL_TV5_A4A4 * (a[4] * a[4]);
}

/* WenoLimiter.h:98-181 compute_weno_coefs<5>.  recon_lo[i][s][ii], recon_hi[s][ii]. */
static void compute_weno_coefs(const double recon_lo[3][3][3], const double recon_hi[5][5], const double u[5],
                               double aw[5], const double idl[4], double sigma) {
  double a_lo[3][3], a_hi[5];
  const double eps = 1.0e-20;
  for (int i = 0; i < 3; i++) {
    for (int ii = 0; ii < 3; ii++) {
      double tmp = 0;
      for (int s = 0; s < 3; s++) tmp += recon_lo[i][s][ii] * u[i + s];
      a_lo[i][ii] = tmp;
    }
  }
  for (int ii = 0; ii < 5; ii++) {
    double tmp = 0;
    for (int s = 0; s < 5; s++) tmp += recon_hi[s][ii] * u[s];
    a_hi[ii] = tmp;
  }
  /* bridge polynomial (WenoLimiter.h:128-136) */
  for (int i = 0; i < 3; i++)
    for (int ii = 0; ii < 3; ii++) a_hi[ii] -= idl[i] * a_lo[i][ii];
  for (int ii = 0; ii < 5; ii++) a_hi[ii] /= idl[3];
  double tv[4];
  for (int i = 0; i < 3; i++) tv[i] = tv3(a_lo[i]);
  tv[3] = tv5(a_hi);
  double lo_avg = 0.0;
  for (int i = 0; i < 3; i++) lo_avg += tv[i];
  lo_avg /= 3;
  tv[3] = lo_avg + (tv[3] - lo_avg) * sigma;
  double wts[4];
  for (int i = 0; i < 4; i++) wts[i] = idl[i] / (tv[i] * tv[i] + eps);
  convexify4(wts);
  map_weights4(idl, wts);
  convexify4(wts);
  for (int i = 0; i < 5; i++) aw[i] = wts[3] * a_hi[i];
  for (int i = 0; i < 3; i++)
    for (int ii = 0; ii < 3; ii++) aw[ii] += wts[i] * a_lo[i][ii];
}

/* Dycore.h:591-604 reconstruct */
static double reconstruct(const double stencil[5], const double c2g[5][2], const double s2c[5][5],
                          const double wrl[3][3][3], const double idl[4], double sigma, int ind) {
  double wc[5];
  compute_weno_coefs(wrl, s2c, stencil, wc, idl, sigma);
  double tmp = 0;
  for (int s = 0; s < 5; s++) tmp += c2g[s][ind] * wc[s];
  return tmp;
}

/* KAT entry point: reconstruct with the constant (uniform-grid) matrices */
double awfl_oracle_reconstruct(const double stencil[5], int ind) {
  const double s2c[5][5] = AWFL_STEN_TO_COEFS_INIT;
  const double wrl[3][3][3] = AWFL_WENO_LOWER_INIT;
  const double c2g[5][2] = AWFL_COEFS_TO_GLL_INIT;
  double idl[4], sigma;
  awfl_oracle_ideal_sigma(idl, &sigma);
  return reconstruct(stencil, c2g, s2c, wrl, idl, sigma, ind);
}

/* full coefficient vector, for convergence-order tests */
void awfl_oracle_weno_coefs(const double stencil[5], double aw[5]) {
  const double s2c[5][5] = AWFL_STEN_TO_COEFS_INIT;
  const double wrl[3][3][3] = AWFL_WENO_LOWER_INIT;
  double idl[4], sigma;
  awfl_oracle_ideal_sigma(idl, &sigma);
  compute_weno_coefs(wrl, s2c, stencil, aw, idl, sigma);
}

/* ---------------------------------------------------------------------------------------------- */
/* yakl::intrinsics::matinv_ge restated (deviation D3): Gauss-Jordan, no pivoting, a[col][row].   */
static void matinv_ge(int n, const double *a, double *inv) {
  double scratch[ORD * ORD];
  for (int icol = 0; icol < n; icol++)
    for (int irow = 0; irow < n; irow++) {
      scratch[icol * n + irow] = a[icol * n + irow];
      inv[icol * n + irow] = (icol == irow) ? 1.0 : 0.0;
    }
  for (int idiag = 0; idiag < n; idiag++) {
    double factor = 1.0 / scratch[idiag * n + idiag];
    for (int icol = idiag; icol < n; icol++) scratch[icol * n + idiag] *= factor;
    for (int icol = 0; icol < n; icol++) inv[icol * n + idiag] *= factor;
    for (int irow = idiag + 1; irow < n; irow++) {
      double f = scratch[idiag * n + irow];
      for (int icol = idiag; icol < n; icol++) scratch[icol * n + irow] -= f * scratch[icol * n + idiag];
      for (int icol = 0; icol < n; icol++) inv[icol * n + irow] -= f * inv[icol * n + idiag];
    }
  }
  for (int idiag = n - 1; idiag >= 1; idiag--) {
    for (int irow = 0; irow < idiag; irow++) {
      double f = scratch[idiag * n + irow];
      for (int icol = irow + 1; icol < n; icol++) scratch[icol * n + irow] -= f * scratch[icol * n + idiag];
      for (int icol = 0; icol < n; icol++) inv[icol * n + irow] -= f * inv[icol * n + idiag];
    }
  }
}

/* TransformMatrices_variable.h:11-32 coefs_to_sten_variable<n>: rslt[i][j] (n x n), locs[n+1] */
static void coefs_to_sten_variable(int n, const double *locs, double *rslt) {
  double locs_pwr[ORD + 1];
  for (int i = 0; i < n + 1; i++) locs_pwr[i] = locs[i];
  for (int i = 0; i < n; i++) rslt[0 * n + i] = 1;
  for (int i = 1; i < n; i++) {
    for (int j = 0; j < n + 1; j++) locs_pwr[j] *= locs[j];
    for (int j = 0; j < n; j++)
      rslt[i * n + j] = 1. / (i + 1.) * (locs_pwr[j] - locs_pwr[j + 1]) / (locs[j] - locs[j + 1]);
  }
}

/* TransformMatrices_variable.h:35-46 sten_to_coefs_variable<n> */
static void sten_to_coefs_variable(int n, const double *locs, double *rslt) {
  double c2s[ORD * ORD];
  coefs_to_sten_variable(n, locs, c2s);
  matinv_ge(n, c2s, rslt);
}

/* TransformMatrices_variable.h:49-69 weno_lower_sten_to_coefs<5>: weno_recon[i][jj][ii] */
static void weno_lower_sten_to_coefs_variable(const double locs[6], double weno_recon[3][3][3]) {
  for (int i = 0; i < 3; i++) {
    double locs_lo[4], recon_lo[9];
    for (int ii = 0; ii < 4; ii++) locs_lo[ii] = locs[i + ii];
    sten_to_coefs_variable(3, locs_lo, recon_lo);
    for (int jj = 0; jj < 3; jj++)
      for (int ii = 0; ii < 3; ii++) weno_recon[i][jj][ii] = recon_lo[jj * 3 + ii];
  }
}

/* KAT entry: variable matrices from 6 edge locations */
void awfl_oracle_variable_matrices(const double locs[6], double s2c[25], double wrl[27]) {
  double w[3][3][3];
  sten_to_coefs_variable(5, locs, s2c);
  weno_lower_sten_to_coefs_variable(locs, w);
  memcpy(wrl, w, sizeof(w));
}

/* ---------------------------------------------------------------------------------------------- */
static double *alloc_nan(size_t n) {
  double *p = (double *)malloc((n ? n : 1) * sizeof(double));
  if (!p) { fprintf(stderr, "awfl_oracle: out of memory (%zu doubles)\n", n); abort(); }
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) p[i] = NAN; /* poison: any stale read shows up as NaN */
  return p;
}

/* Dycore::init (Dycore.h:835-984) minus the idealised-data branch.  consts = {R_d,cp_d,R_v,cp_v,p0,grav}
 * or NULL for the defaults of Dycore.h:871-876.  dz is (nz,nens). */
awfl_oracle_t *awfl_oracle_create(int nens, int nx, int ny, int nz, int nt, double xlen, double ylen,
                                  const double *dz, const unsigned char *positive,
                                  const unsigned char *adds_mass, int idWV, const double *consts) {
  if (nt > MAX_TRACERS || nt < 1) return NULL;
  awfl_oracle_t *o = (awfl_oracle_t *)calloc(1, sizeof(awfl_oracle_t));
  o->nens = nens; o->nx = nx; o->ny = ny; o->nz = nz; o->nt = nt; o->xlen = xlen; o->ylen = ylen;
  o->grav_balance = 1; /* Dycore.h:866 */
  o->R_d = consts ? consts[0] : 287.;
  o->cp_d = consts ? consts[1] : 1003.;
  o->R_v = consts ? consts[2] : 461.;
  o->cp_v = consts ? consts[3] : 1859;
  o->p0 = consts ? consts[4] : 1.e5;
  o->grav = consts ? consts[5] : 9.81;
  o->cv_d = o->cp_d - o->R_d;
  o->gamma_d = o->cp_d / o->cv_d;
  o->kappa_d = o->R_d / o->cp_d;
  o->cv_v = o->R_v - o->cp_v;
  o->C0 = pow(o->R_d * pow(o->p0, -o->kappa_d), o->gamma_d);
  o->idWV = idWV;
  for (int t = 0; t < nt; t++) { o->tracer_positive[t] = positive[t]; o->tracer_adds_mass[t] = adds_mass[t]; }
  {
    const double s2c[5][5] = AWFL_STEN_TO_COEFS_INIT;
    const double wrl[3][3][3] = AWFL_WENO_LOWER_INIT;
    const double c2g[5][2] = AWFL_COEFS_TO_GLL_INIT;
    memcpy(o->s2c, s2c, sizeof(s2c)); memcpy(o->wrl, wrl, sizeof(wrl)); memcpy(o->c2g, c2g, sizeof(c2g));
    awfl_oracle_ideal_sigma(o->idl, &o->sigma);
  }
  size_t nzn = (size_t)nz * nens;
  o->dz = alloc_nan(nzn);
  memcpy(o->dz, dz, nzn * sizeof(double));
  o->vert_s2c = alloc_nan((size_t)(nz + 2) * 25 * nens);
  o->vert_wrl = alloc_nan((size_t)(nz + 2) * 27 * nens);
  o->grav_var = alloc_nan(nzn);
  o->hy_dens_cells = alloc_nan(nzn);
  o->hy_pres_cells = alloc_nan(nzn);
  /* Dycore.h:904-937: per-level vertical matrices (note the off-centre indexing) */
  for (int k = 0; k < nz + 2; k++) {
    for (int iens = 0; iens < nens; iens++) {
      double dzloc[ORD];
      for (int kk = 0; kk < ORD; kk++) {
        int ind1 = -1 + k + kk; if (ind1 < 0) ind1 = 0; if (ind1 > nz - 1) ind1 = nz - 1;
        int ind2 = -1 + k;      if (ind2 < 0) ind2 = 0; if (ind2 > nz - 1) ind2 = nz - 1;
        dzloc[kk] = dz[(size_t)ind1 * nens + iens] / dz[(size_t)ind2 * nens + iens];
      }
      double locs[ORD + 1];
      locs[0] = 0;
      for (int kk = 1; kk < ORD + 1; kk++) locs[kk] = locs[kk - 1] + dzloc[kk - 1];
      double midloc = (locs[(ORD - 1) / 2] + locs[(ORD + 1) / 2]) / 2;
      for (int kk = 0; kk < ORD + 1; kk++) locs[kk] = locs[kk] - midloc;
      double s2c_var[25], wrl_var[3][3][3];
      sten_to_coefs_variable(5, locs, s2c_var);
      weno_lower_sten_to_coefs_variable(locs, wrl_var);
      for (int jj = 0; jj < 5; jj++)
        for (int ii = 0; ii < 5; ii++)
          o->vert_s2c[(((size_t)k * 5 + jj) * 5 + ii) * nens + iens] = s2c_var[jj * 5 + ii];
      for (int kk = 0; kk < 3; kk++)
        for (int jj = 0; jj < 3; jj++)
          for (int ii = 0; ii < 3; ii++)
            o->vert_wrl[((((size_t)k * 3 + kk) * 3 + jj) * 3 + ii) * nens + iens] = wrl_var[kk][jj][ii];
    }
  }
  return o;
}

void awfl_oracle_destroy(awfl_oracle_t *o) {
  if (!o) return;
  free(o->dz); free(o->vert_s2c); free(o->vert_wrl); free(o->grav_var); free(o->hy_dens_cells);
  free(o->hy_pres_cells); free(o);
}

void awfl_oracle_set_grav_balance(awfl_oracle_t *o, int flag) { o->grav_balance = flag; }
double awfl_oracle_get_option(const awfl_oracle_t *o, const char *key) {
  if (!strcmp(key, "R_d")) return o->R_d;
  if (!strcmp(key, "R_v")) return o->R_v;
  if (!strcmp(key, "cp_d")) return o->cp_d;
  if (!strcmp(key, "cp_v")) return o->cp_v;
  if (!strcmp(key, "p0")) return o->p0;
  if (!strcmp(key, "grav")) return o->grav;
  if (!strcmp(key, "cv_d")) return o->cv_d;
  if (!strcmp(key, "cv_v")) return o->cv_v;
  if (!strcmp(key, "gamma_d")) return o->gamma_d;
  if (!strcmp(key, "kappa_d")) return o->kappa_d;
  if (!strcmp(key, "C0")) return o->C0;
  return NAN;
}
const double *awfl_oracle_variable_gravity(const awfl_oracle_t *o) { return o->grav_var; }
const double *awfl_oracle_hy_dens_cells(const awfl_oracle_t *o) { return o->hy_dens_cells; }
const double *awfl_oracle_hy_pressure_cells(const awfl_oracle_t *o) { return o->hy_pres_cells; }
const double *awfl_oracle_vert_sten_to_coefs(const awfl_oracle_t *o) { return o->vert_s2c; }
const double *awfl_oracle_vert_weno_recon_lower(const awfl_oracle_t *o) { return o->vert_wrl; }
void awfl_oracle_set_flux_taps(awfl_oracle_t *o, double *fx, double *fy, double *fz) {
  o->tap_flux_x = fx; o->tap_flux_y = fy; o->tap_flux_z = fz;
}

/* index helpers: halo'd 5-D (l,k,j,i,iens) and 4-D arrays, nens fastest (yakl::styleC) */
#define NZH (nz + 2 * HS)
#define NYH (ny + 2 * HS)
#define NXH (nx + 2 * HS)
#define H5(l, k, j, i, e) ((((((size_t)(l)) * NZH + (k)) * NYH + (j)) * NXH + (i)) * nens + (e))
#define H4(k, j, i, e) (((((size_t)(k)) * NYH + (j)) * NXH + (i)) * nens + (e))
#define C4(k, j, i, e) (((((size_t)(k)) * ny + (j)) * nx + (i)) * nens + (e))
#define C5(l, k, j, i, e) ((((((size_t)(l)) * nz + (k)) * ny + (j)) * nx + (i)) * nens + (e))
#define FX(l, k, j, i, e) ((((((size_t)(l)) * nz + (k)) * ny + (j)) * (nx + 1) + (i)) * nens + (e))
#define FY(l, k, j, i, e) ((((((size_t)(l)) * nz + (k)) * (ny + 1) + (j)) * nx + (i)) * nens + (e))
#define FZ(l, k, j, i, e) ((((((size_t)(l)) * (nz + 1) + (k)) * ny + (j)) * nx + (i)) * nens + (e))
#define DZ(k, e) (o->dz[(size_t)(k) * nens + (e)])

size_t awfl_oracle_halo_elems(const awfl_oracle_t *o) {
  return (size_t)(o->nz + 2 * HS) * (o->ny + 2 * HS) * (o->nx + 2 * HS) * o->nens;
}

/* Dycore.h:1336-1388 convert_coupler_to_dynamics.  tracers_c is (nt,nz,ny,nx,nens). */
void awfl_oracle_convert_coupler_to_dynamics(const awfl_oracle_t *o, const double *rho_d_c, const double *u_c,
                                             const double *v_c, const double *w_c, const double *temp_c,
                                             const double *tracers_c, double *state, double *tracers) {
  const int nens = o->nens, nx = o->nx, ny = o->ny, nz = o->nz, nt = o->nt;
  const double R_d = o->R_d, R_v = o->R_v, gamma_d = o->gamma_d, C0 = o->C0;
#pragma omp parallel for schedule(static)
  for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++)
    for (int e = 0; e < nens; e++) {
      double rho_d = rho_d_c[C4(k, j, i, e)];
      double u = u_c[C4(k, j, i, e)], v = v_c[C4(k, j, i, e)], w = w_c[C4(k, j, i, e)];
      double temp = temp_c[C4(k, j, i, e)];
      double rho_v = tracers_c[C5(o->idWV, k, j, i, e)];
      double press = rho_d * R_d * temp + rho_v * R_v * temp;
      double rho = rho_d;
      for (int tr = 0; tr < nt; tr++) if (o->tracer_adds_mass[tr]) rho += tracers_c[C5(tr, k, j, i, e)];
      double theta = pow(press / C0, 1.0 / gamma_d) / rho;
      state[H5(ID_R, HS + k, HS + j, HS + i, e)] = rho;
      state[H5(ID_U, HS + k, HS + j, HS + i, e)] = rho * u;
      state[H5(ID_V, HS + k, HS + j, HS + i, e)] = rho * v;
      state[H5(ID_W, HS + k, HS + j, HS + i, e)] = rho * w;
      state[H5(ID_T, HS + k, HS + j, HS + i, e)] = rho * theta;
      for (int tr = 0; tr < nt; tr++) tracers[H5(tr, HS + k, HS + j, HS + i, e)] = tracers_c[C5(tr, k, j, i, e)];
    }
}

/* Dycore.h:1281-1331 convert_dynamics_to_coupler */
void awfl_oracle_convert_dynamics_to_coupler(const awfl_oracle_t *o, const double *state, const double *tracers,
                                             double *rho_d_c, double *u_c, double *v_c, double *w_c,
                                             double *temp_c, double *tracers_c) {
  const int nens = o->nens, nx = o->nx, ny = o->ny, nz = o->nz, nt = o->nt;
  const double R_d = o->R_d, R_v = o->R_v, gamma = o->gamma_d, C0 = o->C0;
#pragma omp parallel for schedule(static)
  for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++)
    for (int e = 0; e < nens; e++) {
      double rho = state[H5(ID_R, HS + k, HS + j, HS + i, e)];
      double u = state[H5(ID_U, HS + k, HS + j, HS + i, e)] / rho;
      double v = state[H5(ID_V, HS + k, HS + j, HS + i, e)] / rho;
      double w = state[H5(ID_W, HS + k, HS + j, HS + i, e)] / rho;
      double theta = state[H5(ID_T, HS + k, HS + j, HS + i, e)] / rho;
      double press = C0 * pow(rho * theta, gamma);
      double rho_v = tracers[H5(o->idWV, HS + k, HS + j, HS + i, e)];
      double rho_d = rho;
      for (int tr = 0; tr < nt; tr++) if (o->tracer_adds_mass[tr]) rho_d -= tracers[H5(tr, HS + k, HS + j, HS + i, e)];
      double temp = press / (rho_d * R_d + rho_v * R_v);
      rho_d_c[C4(k, j, i, e)] = rho_d;
      u_c[C4(k, j, i, e)] = u; v_c[C4(k, j, i, e)] = v; w_c[C4(k, j, i, e)] = w;
      temp_c[C4(k, j, i, e)] = temp;
      for (int tr = 0; tr < nt; tr++) tracers_c[C5(tr, k, j, i, e)] = tracers[H5(tr, HS + k, HS + j, HS + i, e)];
    }
}

/* Dycore.h:65-102 compute_time_step */
double awfl_oracle_compute_time_step(const awfl_oracle_t *o, const double *rho_d_c, const double *u_c,
                                     const double *v_c, const double *w_c, const double *temp_c,
                                     const double *tracers_c, double cfl) {
  const int nens = o->nens, nx = o->nx, ny = o->ny, nz = o->nz;
  const double dx = o->xlen / nx, dy = o->ylen / ny;
  double dtmin = INFINITY;
  for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++)
    for (int e = 0; e < nens; e++) {
      double rho_d = rho_d_c[C4(k, j, i, e)];
      double u = u_c[C4(k, j, i, e)], v = v_c[C4(k, j, i, e)], w = w_c[C4(k, j, i, e)];
      double temp = temp_c[C4(k, j, i, e)];
      double rho_v = tracers_c[C5(o->idWV, k, j, i, e)]; /* "water_vapor" by name, Dycore.h:83 */
      double rho = rho_d + rho_v;
      double p = (rho_d * o->R_d + rho_v * o->R_v) * temp;
      double cs = sqrt(o->gamma_d * p / rho);
      double dtx = cfl * dx / (fabs(u) + cs);
      double dty = cfl * dy / (fabs(v) + cs);
      double dtz = cfl * DZ(k, e) / (fabs(w) + cs);
      double d = fmin(fmin(dtx, dty), dtz);
      if (d < dtmin) dtmin = d;
    }
  return dtmin;
}

/* Dycore.h:608-711 halo_exchange, with deviation D1 in the vertical-ghost kernel */
static void halo_exchange(const awfl_oracle_t *o, double *state, double *tracers, double *pressure) {
  const int nens = o->nens, nx = o->nx, ny = o->ny, nz = o->nz, nt = o->nt;
  const int sim2d = (ny == 1);
  const double gamma = o->gamma_d, C0 = o->C0, grav = o->grav;
  const int npack = NUM_STATE + nt + 1;
  for (int v = 0; v < npack; v++) for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++)
    for (int ii = 0; ii < HS; ii++) for (int e = 0; e < nens; e++) {
      if (v < NUM_STATE) {
        state[H5(v, HS + k, HS + j, nx + HS + ii, e)] = state[H5(v, HS + k, HS + j, HS + ii, e)];
        state[H5(v, HS + k, HS + j, ii, e)] = state[H5(v, HS + k, HS + j, nx + ii, e)];
      } else if (v < NUM_STATE + nt) {
        tracers[H5(v - NUM_STATE, HS + k, HS + j, nx + HS + ii, e)] = tracers[H5(v - NUM_STATE, HS + k, HS + j, HS + ii, e)];
        tracers[H5(v - NUM_STATE, HS + k, HS + j, ii, e)] = tracers[H5(v - NUM_STATE, HS + k, HS + j, nx + ii, e)];
      } else {
        pressure[H4(HS + k, HS + j, nx + HS + ii, e)] = pressure[H4(HS + k, HS + j, HS + ii, e)];
        pressure[H4(HS + k, HS + j, ii, e)] = pressure[H4(HS + k, HS + j, nx + ii, e)];
      }
    }
  if (!sim2d) {
    for (int v = 0; v < npack; v++) for (int k = 0; k < nz; k++) for (int jj = 0; jj < HS; jj++)
      for (int i = 0; i < nx; i++) for (int e = 0; e < nens; e++) {
        if (v < NUM_STATE) {
          state[H5(v, HS + k, ny + HS + jj, HS + i, e)] = state[H5(v, HS + k, HS + jj, HS + i, e)];
          state[H5(v, HS + k, jj, HS + i, e)] = state[H5(v, HS + k, ny + jj, HS + i, e)];
        } else if (v < NUM_STATE + nt) {
          tracers[H5(v - NUM_STATE, HS + k, ny + HS + jj, HS + i, e)] = tracers[H5(v - NUM_STATE, HS + k, HS + jj, HS + i, e)];
          tracers[H5(v - NUM_STATE, HS + k, jj, HS + i, e)] = tracers[H5(v - NUM_STATE, HS + k, ny + jj, HS + i, e)];
        } else {
          pressure[H4(HS + k, ny + HS + jj, HS + i, e)] = pressure[H4(HS + k, HS + jj, HS + i, e)];
          pressure[H4(HS + k, jj, HS + i, e)] = pressure[H4(HS + k, ny + jj, HS + i, e)];
        }
      }
  }
  /* vertical boundary conditions, Dycore.h:662-710.  D1: first all copies, then density/pressure. */
  for (int kk = 0; kk < HS; kk++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++)
    for (int e = 0; e < nens; e++) {
      for (int l = 1; l < NUM_STATE; l++) {
        if (l == ID_W) {
          state[H5(l, kk, HS + j, HS + i, e)] = 0;
          state[H5(l, HS + nz + kk, HS + j, HS + i, e)] = 0;
        } else {
          state[H5(l, kk, HS + j, HS + i, e)] = state[H5(l, HS + 0, HS + j, HS + i, e)];
          state[H5(l, HS + nz + kk, HS + j, HS + i, e)] = state[H5(l, HS + nz - 1, HS + j, HS + i, e)];
        }
      }
      for (int l = 0; l < nt; l++) {
        tracers[H5(l, kk, HS + j, HS + i, e)] = tracers[H5(l, HS + 0, HS + j, HS + i, e)];
        tracers[H5(l, HS + nz + kk, HS + j, HS + i, e)] = tracers[H5(l, HS + nz - 1, HS + j, HS + i, e)];
      }
      if (!o->grav_balance) {
        pressure[H4(kk, HS + j, HS + i, e)] = pressure[H4(HS + 0, HS + j, HS + i, e)];
        pressure[H4(HS + nz + kk, HS + j, HS + i, e)] = pressure[H4(HS + nz - 1, HS + j, HS + i, e)];
      }
    }
  for (int kk = 0; kk < HS; kk++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++)
    for (int e = 0; e < nens; e++) {
      {
        int k0 = HS, k = k0 - 1 - kk;
        double rho0 = state[H5(ID_R, k0, HS + j, HS + i, e)];
        double theta0 = state[H5(ID_T, k0, HS + j, HS + i, e)];
        double rho0_gm1 = pow(rho0, gamma - 1);
        double theta0_g = pow(theta0, gamma);
        state[H5(ID_R, k, HS + j, HS + i, e)] =
            pow(rho0_gm1 + grav * (gamma - 1) * DZ(k0 - HS, e) * (kk + 1) / (gamma * C0 * theta0_g), 1.0 / (gamma - 1));
        if (o->grav_balance) {
          double rt = state[H5(ID_R, k, HS + j, HS + i, e)] * state[H5(ID_T, k, HS + j, HS + i, e)];
          pressure[H4(k, HS + j, HS + i, e)] = C0 * pow(rt, gamma);
        }
      }
      {
        int k0 = HS + nz - 1, k = k0 + 1 + kk;
        double rho0 = state[H5(ID_R, k0, HS + j, HS + i, e)];
        double theta0 = state[H5(ID_T, k0, HS + j, HS + i, e)];
        double rho0_gm1 = pow(rho0, gamma - 1);
        double theta0_g = pow(theta0, gamma);
        state[H5(ID_R, k, HS + j, HS + i, e)] =
            pow(rho0_gm1 - grav * (gamma - 1) * DZ(k0 - HS, e) * (kk + 1) / (gamma * C0 * theta0_g), 1.0 / (gamma - 1));
        if (o->grav_balance) {
          double rt = state[H5(ID_R, k, HS + j, HS + i, e)] * state[H5(ID_T, k, HS + j, HS + i, e)];
          pressure[H4(k, HS + j, HS + i, e)] = C0 * pow(rt, gamma);
        }
      }
    }
}

static void load_vert(const awfl_oracle_t *o, int k, int e, double s2c[5][5], double wrl[3][3][3]) {
  const int nens = o->nens;
  for (int i1 = 0; i1 < 5; i1++)
    for (int i2 = 0; i2 < 5; i2++) s2c[i1][i2] = o->vert_s2c[(((size_t)k * 5 + i1) * 5 + i2) * nens + e];
  for (int i1 = 0; i1 < 3; i1++)
    for (int i2 = 0; i2 < 3; i2++)
      for (int i3 = 0; i3 < 3; i3++)
        wrl[i1][i2][i3] = o->vert_wrl[((((size_t)k * 3 + i1) * 3 + i2) * 3 + i3) * nens + e];
}

/* KAT entry point for the vertical direction: reconstruct with the per-level matrices of index k (member e), as the z sweep
 * does (Dycore.h:454-469: the face-k "L" stencil uses vert_*(k), the "R" stencil vert_*(k+1)); ind as awfl_oracle_reconstruct */
double awfl_oracle_reconstruct_level(const awfl_oracle_t *o, int k, int e, const double stencil[5], int ind) {
  const double c2g[5][2] = AWFL_COEFS_TO_GLL_INIT;
  double s2c[5][5], wrl[3][3][3], idl[4], sigma;
  load_vert(o, k, e, s2c, wrl);
  awfl_oracle_ideal_sigma(idl, &sigma);
  return reconstruct(stencil, c2g, s2c, wrl, idl, sigma, ind);
}

/* Dycore.h:262-586 compute_tendencies.  state/tracers are halo'd and modified in place exactly as the
 * reference does (divide by rho, multiply back).  tracers_tend carries the FCT mass seed in, tendencies out. */
void awfl_oracle_compute_tendencies(const awfl_oracle_t *o, double *state, double *state_tend, double *tracers,
                                    double *tracers_tend, double dt) {
  const int nens = o->nens, nx = o->nx, ny = o->ny, nz = o->nz, nt = o->nt;
  const double dx = o->xlen / nx, dy = o->ylen / ny;
  const int sim2d = (ny == 1);
  const double C0 = o->C0, gamma_d = o->gamma_d, grav = o->grav;
  const int grav_balance = o->grav_balance;
  const double (*s2c)[5] = o->s2c;
  const double (*c2g)[2] = o->c2g;
  const double (*wrl)[3][3] = o->wrl;
  const double *idl = o->idl;
  const double sigma = o->sigma;

  double *pressure = alloc_nan(awfl_oracle_halo_elems(o));
  /* Dycore.h:310-321 */
#pragma omp parallel for schedule(static)
  for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++)
    for (int e = 0; e < nens; e++) {
      if (grav_balance) {
        pressure[H4(HS + k, HS + j, HS + i, e)] = C0 * pow(state[H5(ID_T, HS + k, HS + j, HS + i, e)], gamma_d);
      } else {
        pressure[H4(HS + k, HS + j, HS + i, e)] =
            C0 * pow(state[H5(ID_T, HS + k, HS + j, HS + i, e)], gamma_d) - o->hy_pres_cells[(size_t)k * nens + e];
      }
      double r = state[H5(ID_R, HS + k, HS + j, HS + i, e)];
      state[H5(ID_U, HS + k, HS + j, HS + i, e)] /= r;
      state[H5(ID_V, HS + k, HS + j, HS + i, e)] /= r;
      state[H5(ID_W, HS + k, HS + j, HS + i, e)] /= r;
      state[H5(ID_T, HS + k, HS + j, HS + i, e)] /= r;
      for (int tr = 0; tr < nt; tr++) tracers[H5(tr, HS + k, HS + j, HS + i, e)] /= r;
    }

  halo_exchange(o, state, tracers, pressure);

  double *sfx = alloc_nan((size_t)NUM_STATE * nz * ny * (nx + 1) * nens);
  double *sfy = alloc_nan((size_t)NUM_STATE * nz * (ny + 1) * nx * nens);
  double *sfz = alloc_nan((size_t)NUM_STATE * (nz + 1) * ny * nx * nens);
  double *tfx = alloc_nan((size_t)nt * nz * ny * (nx + 1) * nens);
  double *tfy = alloc_nan((size_t)nt * nz * (ny + 1) * nx * nens);
  double *tfz = alloc_nan((size_t)nt * (nz + 1) * ny * nx * nens);

  /* Dycore.h:334-519 */
#pragma omp parallel for collapse(2) schedule(static)
  for (int k = 0; k < nz + 1; k++) for (int j = 0; j < ny + 1; j++) for (int i = 0; i < nx + 1; i++)
    for (int e = 0; e < nens; e++) {
      const double cs = 350;
      double stencil[5];
      /* X */
      if (j < ny && k < nz) {
        double ru, pp;
        {
          int i_upw = 0;
          for (int s = 0; s < ORD; s++) stencil[s] = state[H5(ID_R, HS + k, HS + j, i + i_upw + s, e)] * state[H5(ID_U, HS + k, HS + j, i + i_upw + s, e)];
          double ru_L = reconstruct(stencil, c2g, s2c, wrl, idl, sigma, 1 - i_upw);
          i_upw = 1;
          for (int s = 0; s < ORD; s++) stencil[s] = state[H5(ID_R, HS + k, HS + j, i + i_upw + s, e)] * state[H5(ID_U, HS + k, HS + j, i + i_upw + s, e)];
          double ru_R = reconstruct(stencil, c2g, s2c, wrl, idl, sigma, 1 - i_upw);
          i_upw = 0;
          for (int s = 0; s < ORD; s++) stencil[s] = pressure[H4(HS + k, HS + j, i + i_upw + s, e)];
          double pp_L = reconstruct(stencil, c2g, s2c, wrl, idl, sigma, 1 - i_upw);
          i_upw = 1;
          for (int s = 0; s < ORD; s++) stencil[s] = pressure[H4(HS + k, HS + j, i + i_upw + s, e)];
          double pp_R = reconstruct(stencil, c2g, s2c, wrl, idl, sigma, 1 - i_upw);
          double w1 = 0.5 * (pp_R - cs * ru_R);
          double w2 = 0.5 * (pp_L + cs * ru_L);
          pp = w1 + w2;
          ru = (w2 - w1) / cs;
          sfx[FX(ID_R, k, j, i, e)] = ru;
        }
        int i_upw = ru > 0 ? 0 : 1;
        for (int l = ID_U; l <= ID_T; l++) {
          for (int s = 0; s < ORD; s++) stencil[s] = state[H5(l, HS + k, HS + j, i + i_upw + s, e)];
          double val = ru * reconstruct(stencil, c2g, s2c, wrl, idl, sigma, 1 - i_upw);
          if (l == ID_U) val = val + pp;
          sfx[FX(l, k, j, i, e)] = val;
        }
        for (int tr = 0; tr < nt; tr++) {
          for (int s = 0; s < ORD; s++) stencil[s] = tracers[H5(tr, HS + k, HS + j, i + i_upw + s, e)];
          tfx[FX(tr, k, j, i, e)] = ru * reconstruct(stencil, c2g, s2c, wrl, idl, sigma, 1 - i_upw);
        }
      }
      /* Y */
      if (i < nx && k < nz) {
        if (!sim2d) {
          double rv, pp;
          {
            int j_upw = 0;
            for (int s = 0; s < ORD; s++) stencil[s] = state[H5(ID_R, HS + k, j + j_upw + s, HS + i, e)] * state[H5(ID_V, HS + k, j + j_upw + s, HS + i, e)];
            double rv_L = reconstruct(stencil, c2g, s2c, wrl, idl, sigma, 1 - j_upw);
            j_upw = 1;
            for (int s = 0; s < ORD; s++) stencil[s] = state[H5(ID_R, HS + k, j + j_upw + s, HS + i, e)] * state[H5(ID_V, HS + k, j + j_upw + s, HS + i, e)];
            double rv_R = reconstruct(stencil, c2g, s2c, wrl, idl, sigma, 1 - j_upw);
            j_upw = 0;
            for (int s = 0; s < ORD; s++) stencil[s] = pressure[H4(HS + k, j + j_upw + s, HS + i, e)];
            double pp_L = reconstruct(stencil, c2g, s2c, wrl, idl, sigma, 1 - j_upw);
            j_upw = 1;
            for (int s = 0; s < ORD; s++) stencil[s] = pressure[H4(HS + k, j + j_upw + s, HS + i, e)];
            double pp_R = reconstruct(stencil, c2g, s2c, wrl, idl, sigma, 1 - j_upw);
            double w1 = 0.5 * (pp_R - cs * rv_R);
            double w2 = 0.5 * (pp_L + cs * rv_L);
            pp = w1 + w2;
            rv = (w2 - w1) / cs;
            sfy[FY(ID_R, k, j, i, e)] = rv;
          }
          int j_upw = rv > 0 ? 0 : 1;
          for (int l = ID_U; l <= ID_T; l++) {
            for (int s = 0; s < ORD; s++) stencil[s] = state[H5(l, HS + k, j + j_upw + s, HS + i, e)];
            double val = rv * reconstruct(stencil, c2g, s2c, wrl, idl, sigma, 1 - j_upw);
            if (l == ID_V) val = val + pp;
            sfy[FY(l, k, j, i, e)] = val;
          }
          for (int tr = 0; tr < nt; tr++) {
            for (int s = 0; s < ORD; s++) stencil[s] = tracers[H5(tr, HS + k, j + j_upw + s, HS + i, e)];
            tfy[FY(tr, k, j, i, e)] = rv * reconstruct(stencil, c2g, s2c, wrl, idl, sigma, 1 - j_upw);
          }
        } else {
          for (int l = 0; l < NUM_STATE; l++) sfy[FY(l, k, j, i, e)] = 0;
          for (int tr = 0; tr < nt; tr++) tfy[FY(tr, k, j, i, e)] = 0;
        }
      }
      /* Z */
      if (i < nx && j < ny) {
        double s2c_loc[2][5][5], wrl_loc[2][3][3][3];
        load_vert(o, k, e, s2c_loc[0], wrl_loc[0]);
        load_vert(o, k + 1, e, s2c_loc[1], wrl_loc[1]);
        double rw, pp;
        {
          int k_upw = 0;
          for (int s = 0; s < ORD; s++) stencil[s] = state[H5(ID_R, k + k_upw + s, HS + j, HS + i, e)] * state[H5(ID_W, k + k_upw + s, HS + j, HS + i, e)];
          double rw_L = reconstruct(stencil, c2g, s2c_loc[k_upw], wrl_loc[k_upw], idl, sigma, 1 - k_upw);
          if (k == 0 || k == nz) rw_L = 0;
          k_upw = 1;
          for (int s = 0; s < ORD; s++) stencil[s] = state[H5(ID_R, k + k_upw + s, HS + j, HS + i, e)] * state[H5(ID_W, k + k_upw + s, HS + j, HS + i, e)];
          double rw_R = reconstruct(stencil, c2g, s2c_loc[k_upw], wrl_loc[k_upw], idl, sigma, 1 - k_upw);
          if (k == 0 || k == nz) rw_R = 0;
          k_upw = 0;
          for (int s = 0; s < ORD; s++) stencil[s] = pressure[H4(k + k_upw + s, HS + j, HS + i, e)];
          double pp_L = reconstruct(stencil, c2g, s2c_loc[k_upw], wrl_loc[k_upw], idl, sigma, 1 - k_upw);
          k_upw = 1;
          for (int s = 0; s < ORD; s++) stencil[s] = pressure[H4(k + k_upw + s, HS + j, HS + i, e)];
          double pp_R = reconstruct(stencil, c2g, s2c_loc[k_upw], wrl_loc[k_upw], idl, sigma, 1 - k_upw);
          double w1 = 0.5 * (pp_R - cs * rw_R);
          double w2 = 0.5 * (pp_L + cs * rw_L);
          pp = w1 + w2;
          rw = (w2 - w1) / cs;
          if (k == 0 || k == nz) rw = 0;
          sfz[FZ(ID_R, k, j, i, e)] = rw;
        }
        int k_upw = rw > 0 ? 0 : 1;
        for (int l = ID_U; l <= ID_T; l++) {
          for (int s = 0; s < ORD; s++) stencil[s] = state[H5(l, k + k_upw + s, HS + j, HS + i, e)];
          double val = rw * reconstruct(stencil, c2g, s2c_loc[k_upw], wrl_loc[k_upw], idl, sigma, 1 - k_upw);
          if (l == ID_W) val = val + pp;
          sfz[FZ(l, k, j, i, e)] = val;
        }
        for (int tr = 0; tr < nt; tr++) {
          for (int s = 0; s < ORD; s++) stencil[s] = tracers[H5(tr, k + k_upw + s, HS + j, HS + i, e)];
          tfz[FZ(tr, k, j, i, e)] = rw * reconstruct(stencil, c2g, s2c_loc[k_upw], wrl_loc[k_upw], idl, sigma, 1 - k_upw);
        }
      }
    }

  /* Dycore.h:525-550: multiply rho back; FCT.  (Order-independent: a face is only ever scaled by the
   * one adjacent cell it flows out of, Dycore.h:521-524.) */
#pragma omp parallel for schedule(static)
  for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++)
    for (int e = 0; e < nens; e++) {
      double r = state[H5(ID_R, HS + k, HS + j, HS + i, e)];
      state[H5(ID_U, HS + k, HS + j, HS + i, e)] *= r;
      state[H5(ID_V, HS + k, HS + j, HS + i, e)] *= r;
      state[H5(ID_W, HS + k, HS + j, HS + i, e)] *= r;
      state[H5(ID_T, HS + k, HS + j, HS + i, e)] *= r;
      for (int tr = 0; tr < nt; tr++) {
        tracers[H5(tr, HS + k, HS + j, HS + i, e)] *= r;
        if (o->tracer_positive[tr]) {
          double dzk = DZ(k, e);
          double mass_available = fmax(tracers_tend[C5(tr, k, j, i, e)], 0.0) * dx * dy * dzk;
          double flux_out_x = (fmax(tfx[FX(tr, k, j, i + 1, e)], 0.0) - fmin(tfx[FX(tr, k, j, i, e)], 0.0)) / dx;
          double flux_out_y = (fmax(tfy[FY(tr, k, j + 1, i, e)], 0.0) - fmin(tfy[FY(tr, k, j, i, e)], 0.0)) / dy;
          double flux_out_z = (fmax(tfz[FZ(tr, k + 1, j, i, e)], 0.0) - fmin(tfz[FZ(tr, k, j, i, e)], 0.0)) / dzk;
          double mass_out = (flux_out_x + flux_out_y + flux_out_z) * dt * dx * dy * dzk;
          if (mass_out > mass_available) {
            double mult = mass_available / mass_out;
            if (tfx[FX(tr, k, j, i + 1, e)] > 0) tfx[FX(tr, k, j, i + 1, e)] *= mult;
            if (tfx[FX(tr, k, j, i, e)] < 0) tfx[FX(tr, k, j, i, e)] *= mult;
            if (tfy[FY(tr, k, j + 1, i, e)] > 0) tfy[FY(tr, k, j + 1, i, e)] *= mult;
            if (tfy[FY(tr, k, j, i, e)] < 0) tfy[FY(tr, k, j, i, e)] *= mult;
            if (tfz[FZ(tr, k + 1, j, i, e)] > 0) tfz[FZ(tr, k + 1, j, i, e)] *= mult;
            if (tfz[FZ(tr, k, j, i, e)] < 0) tfz[FZ(tr, k, j, i, e)] *= mult;
          }
        }
      }
    }

  /* Dycore.h:553-584 */
#pragma omp parallel for schedule(static)
  for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++)
    for (int e = 0; e < nens; e++) {
      for (int l = 0; l < NUM_STATE; l++) {
        double t = -(sfx[FX(l, k, j, i + 1, e)] - sfx[FX(l, k, j, i, e)]) / dx
                   - (sfy[FY(l, k, j + 1, i, e)] - sfy[FY(l, k, j, i, e)]) / dy
                   - (sfz[FZ(l, k + 1, j, i, e)] - sfz[FZ(l, k, j, i, e)]) / DZ(k, e);
        if (l == ID_W) {
          if (grav_balance) t += -o->grav_var[(size_t)k * nens + e] * state[H5(ID_R, HS + k, HS + j, HS + i, e)];
          else t += -grav * (state[H5(ID_R, HS + k, HS + j, HS + i, e)] - o->hy_dens_cells[(size_t)k * nens + e]);
        }
        if (l == ID_V && sim2d) t = 0;
        state_tend[C5(l, k, j, i, e)] = t;
      }
      for (int l = 0; l < nt; l++) {
        double fx = tfx[FX(l, k, j, i, e)], fxp1 = tfx[FX(l, k, j, i + 1, e)];
        double fy = tfy[FY(l, k, j, i, e)], fyp1 = tfy[FY(l, k, j + 1, i, e)];
        double fz = tfz[FZ(l, k, j, i, e)], fzp1 = tfz[FZ(l, k + 1, j, i, e)];
        if (i == 0) fx = fmin(fx, tfx[FX(l, k, j, nx, e)]);
        if (i == nx - 1) fxp1 = fmin(fxp1, tfx[FX(l, k, j, 0, e)]);
        if (j == 0) fy = fmin(fy, tfy[FY(l, k, ny, i, e)]);
        if (j == ny - 1) fyp1 = fmin(fyp1, tfy[FY(l, k, 0, i, e)]);
        tracers_tend[C5(l, k, j, i, e)] = -(fxp1 - fx) / dx - (fyp1 - fy) / dy - (fzp1 - fz) / DZ(k, e);
      }
    }

  if (o->tap_flux_x) { /* debug taps: [state 5 | tracers nt] x faces, post-FCT */
    size_t nsx = (size_t)NUM_STATE * nz * ny * (nx + 1) * nens, ntx = (size_t)nt * nz * ny * (nx + 1) * nens;
    size_t nsy = (size_t)NUM_STATE * nz * (ny + 1) * nx * nens, nty = (size_t)nt * nz * (ny + 1) * nx * nens;
    size_t nsz = (size_t)NUM_STATE * (nz + 1) * ny * nx * nens, ntz = (size_t)nt * (nz + 1) * ny * nx * nens;
    memcpy(o->tap_flux_x, sfx, nsx * 8); memcpy(o->tap_flux_x + nsx, tfx, ntx * 8);
    memcpy(o->tap_flux_y, sfy, nsy * 8); memcpy(o->tap_flux_y + nsy, tfy, nty * 8);
    memcpy(o->tap_flux_z, sfz, nsz * 8); memcpy(o->tap_flux_z + nsz, tfz, ntz * 8);
  }
  free(pressure); free(sfx); free(sfy); free(sfz); free(tfx); free(tfy); free(tfz);
}

/* Dycore.h:1392-1504 declare_current_profile_as_hydrostatic (use_gcm_data=false path; the gcm path takes
 * gcm columns (nz,nens) when gcm != NULL: {rho_d, temp, rho_v, rho_c, rho_i}). */
void awfl_oracle_declare_hydrostatic(awfl_oracle_t *o, const double *rho_d_c, const double *u_c, const double *v_c,
                                     const double *w_c, const double *temp_c, const double *tracers_c,
                                     const double *const *gcm) {
  const int nens = o->nens, nx = o->nx, ny = o->ny, nz = o->nz, nt = o->nt;
  const double C0 = o->C0, gamma_d = o->gamma_d;
  size_t nh = awfl_oracle_halo_elems(o);
  double *state = alloc_nan(NUM_STATE * nh), *tracers = alloc_nan((size_t)nt * nh);
  if (gcm) { /* Dycore.h:1415-1434 */
    for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++)
      for (int e = 0; e < nens; e++) {
        size_t c = (size_t)k * nens + e;
        double rho_d = gcm[0][c], rho_v = gcm[2][c];
        double rho = gcm[0][c] + gcm[2][c] + gcm[3][c] + gcm[4][c];
        double temp = gcm[1][c];
        double p = (rho_d * o->R_d + rho_v * o->R_v) * temp;
        double rho_theta = pow(p / C0, 1.0 / gamma_d);
        state[H5(ID_R, HS + k, HS + j, HS + i, e)] = rho;
        state[H5(ID_U, HS + k, HS + j, HS + i, e)] = 0;
        state[H5(ID_V, HS + k, HS + j, HS + i, e)] = 0;
        state[H5(ID_W, HS + k, HS + j, HS + i, e)] = 0;
        state[H5(ID_T, HS + k, HS + j, HS + i, e)] = rho_theta;
        for (int tr = 0; tr < nt; tr++) tracers[H5(tr, HS + k, HS + j, HS + i, e)] = 0;
      }
  } else {
    awfl_oracle_convert_coupler_to_dynamics(o, rho_d_c, u_c, v_c, w_c, temp_c, tracers_c, state, tracers);
  }
  if (o->grav_balance) {
    double *pressure = alloc_nan(nh);
    for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++)
      for (int e = 0; e < nens; e++) {
        pressure[H4(HS + k, HS + j, HS + i, e)] = C0 * pow(state[H5(ID_T, HS + k, HS + j, HS + i, e)], gamma_d);
        state[H5(ID_T, HS + k, HS + j, HS + i, e)] /= state[H5(ID_R, HS + k, HS + j, HS + i, e)];
        if (j == 0 && i == 0) o->grav_var[(size_t)k * nens + e] = 0;
      }
    halo_exchange(o, state, tracers, pressure);
    double *pint = alloc_nan((size_t)(nz + 1) * ny * nx * nens);
    for (int k = 0; k < nz + 1; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++)
      for (int e = 0; e < nens; e++) {
        double s2c_loc[2][5][5], wrl_loc[2][3][3][3], stencil[5];
        load_vert(o, k, e, s2c_loc[0], wrl_loc[0]);
        load_vert(o, k + 1, e, s2c_loc[1], wrl_loc[1]);
        int k_upw = 0;
        for (int s = 0; s < ORD; s++) stencil[s] = pressure[H4(k + k_upw + s, HS + j, HS + i, e)];
        double pp_L = reconstruct(stencil, o->c2g, s2c_loc[k_upw], wrl_loc[k_upw], o->idl, o->sigma, 1 - k_upw);
        k_upw = 1;
        for (int s = 0; s < ORD; s++) stencil[s] = pressure[H4(k + k_upw + s, HS + j, HS + i, e)];
        double pp_R = reconstruct(stencil, o->c2g, s2c_loc[k_upw], wrl_loc[k_upw], o->idl, o->sigma, 1 - k_upw);
        pint[(((size_t)k * ny + j) * nx + i) * nens + e] = 0.5 * (pp_L + pp_R);
      }
    double r_nx_ny = 1. / (nx * ny);
    /* serial atomicAdd order: k, j, i, iens nested loops (first bound slowest) */
    for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++)
      for (int e = 0; e < nens; e++) {
        double dens = state[H5(ID_R, HS + k, HS + j, HS + i, e)];
        double pu = pint[(((size_t)(k + 1) * ny + j) * nx + i) * nens + e];
        double pl = pint[(((size_t)k * ny + j) * nx + i) * nens + e];
        o->grav_var[(size_t)k * nens + e] += -(pu - pl) / (dens * DZ(k, e)) * r_nx_ny;
      }
    free(pint); free(pressure);
  } else {
    double r_nx_ny = 1. / (nx * ny);
    for (size_t c = 0; c < (size_t)nz * nens; c++) { o->hy_dens_cells[c] = 0; o->hy_pres_cells[c] = 0; }
    for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++)
      for (int e = 0; e < nens; e++) {
        double press = C0 * pow(state[H5(ID_T, HS + k, HS + j, HS + i, e)], gamma_d);
        o->hy_pres_cells[(size_t)k * nens + e] += press * r_nx_ny;
        o->hy_dens_cells[(size_t)k * nens + e] += state[H5(ID_R, HS + k, HS + j, HS + i, e)] * r_nx_ny;
      }
  }
  free(state); free(tracers);
}

/* Dycore.h:107-255 timeStep.  Coupler arrays are updated in place.  dt_dyn_in > 0 overrides the CFL time
 * step (used to impose the ensemble-global minimum when the ensemble is sharded).  Returns ncycles. */
int awfl_oracle_time_step(awfl_oracle_t *o, double *rho_d_c, double *u_c, double *v_c, double *w_c, double *temp_c,
                          double *tracers_c, double dt_phys, double dt_dyn_in, double *dt_dyn_out) {
  const int nens = o->nens, nx = o->nx, ny = o->ny, nz = o->nz, nt = o->nt;
  size_t nh = awfl_oracle_halo_elems(o);
  size_t nc = (size_t)nz * ny * nx * nens;
  double *state = alloc_nan(NUM_STATE * nh), *tracers = alloc_nan((size_t)nt * nh);
  awfl_oracle_convert_coupler_to_dynamics(o, rho_d_c, u_c, v_c, w_c, temp_c, tracers_c, state, tracers);
  /* Dycore.h:130-134 */
  for (int l = 0; l < nt; l++) for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++)
    for (int e = 0; e < nens; e++)
      if (o->tracer_positive[l]) tracers[H5(l, HS + k, HS + j, HS + i, e)] = fmax(0.0, tracers[H5(l, HS + k, HS + j, HS + i, e)]);
  double dt_dyn = dt_dyn_in > 0 ? dt_dyn_in
                                : awfl_oracle_compute_time_step(o, rho_d_c, u_c, v_c, w_c, temp_c, tracers_c, 0.8);
  int ncycles = (int)ceil(dt_phys / dt_dyn);
  dt_dyn = dt_phys / ncycles;
  if (dt_dyn_out) *dt_dyn_out = dt_dyn;
  for (int icycle = 0; icycle < ncycles; icycle++) {
    double *state_tmp = alloc_nan(NUM_STATE * nh), *state_tend = alloc_nan(NUM_STATE * nc);
    double *tracers_tmp = alloc_nan((size_t)nt * nh), *tracers_tend = alloc_nan((size_t)nt * nc);
    /* stage 1, Dycore.h:156-176 */
    for (int l = 0; l < nt; l++) for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++)
      for (int e = 0; e < nens; e++) tracers_tend[C5(l, k, j, i, e)] = tracers[H5(l, HS + k, HS + j, HS + i, e)];
    awfl_oracle_compute_tendencies(o, state, state_tend, tracers, tracers_tend, dt_dyn);
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) for (int e = 0; e < nens; e++) {
      for (int l = 0; l < NUM_STATE; l++)
        state_tmp[H5(l, HS + k, HS + j, HS + i, e)] = state[H5(l, HS + k, HS + j, HS + i, e)] + dt_dyn * state_tend[C5(l, k, j, i, e)];
      for (int l = 0; l < nt; l++) {
        size_t h = H5(l, HS + k, HS + j, HS + i, e);
        tracers_tmp[h] = tracers[h] + dt_dyn * tracers_tend[C5(l, k, j, i, e)];
        if (o->tracer_positive[l]) tracers_tmp[h] = fmax(0.0, tracers_tmp[h]);
        tracers_tend[C5(l, k, j, i, e)] = (3.0 / 4.0) * tracers[h] + (1.0 / 4.0) * tracers_tmp[h];
      }
    }
    /* stage 2, Dycore.h:180-200 */
    awfl_oracle_compute_tendencies(o, state_tmp, state_tend, tracers_tmp, tracers_tend, (1.0 / 4.0) * dt_dyn);
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) for (int e = 0; e < nens; e++) {
      for (int l = 0; l < NUM_STATE; l++) {
        size_t h = H5(l, HS + k, HS + j, HS + i, e);
        state_tmp[h] = (3.0 / 4.0) * state[h] + (1.0 / 4.0) * state_tmp[h] + (1.0 / 4.0) * dt_dyn * state_tend[C5(l, k, j, i, e)];
      }
      for (int l = 0; l < nt; l++) {
        size_t h = H5(l, HS + k, HS + j, HS + i, e);
        tracers_tmp[h] = (3.0 / 4.0) * tracers[h] + (1.0 / 4.0) * tracers_tmp[h] + (1.0 / 4.0) * dt_dyn * tracers_tend[C5(l, k, j, i, e)];
        if (o->tracer_positive[l]) tracers_tmp[h] = fmax(0.0, tracers_tmp[h]);
        tracers_tend[C5(l, k, j, i, e)] = (1.0 / 3.0) * tracers[h] + (2.0 / 3.0) * tracers_tmp[h];
      }
    }
    /* stage 3, Dycore.h:204-221 */
    awfl_oracle_compute_tendencies(o, state_tmp, state_tend, tracers_tmp, tracers_tend, (2.0 / 3.0) * dt_dyn);
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) for (int e = 0; e < nens; e++) {
      for (int l = 0; l < NUM_STATE; l++) {
        size_t h = H5(l, HS + k, HS + j, HS + i, e);
        state[h] = (1.0 / 3.0) * state[h] + (2.0 / 3.0) * state_tmp[h] + (2.0 / 3.0) * dt_dyn * state_tend[C5(l, k, j, i, e)];
      }
      for (int l = 0; l < nt; l++) {
        size_t h = H5(l, HS + k, HS + j, HS + i, e);
        tracers[h] = (1.0 / 3.0) * tracers[h] + (2.0 / 3.0) * tracers_tmp[h] + (2.0 / 3.0) * dt_dyn * tracers_tend[C5(l, k, j, i, e)];
        if (o->tracer_positive[l]) tracers[h] = fmax(0.0, tracers[h]);
      }
    }
    free(state_tmp); free(state_tend); free(tracers_tmp); free(tracers_tend);
  }
  awfl_oracle_convert_dynamics_to_coupler(o, state, tracers, rho_d_c, u_c, v_c, w_c, temp_c, tracers_c);
  free(state); free(tracers);
  return ncycles;
}

/* ---------------------------------------------------------------------------------------------- */
/* modules::sponge_layer (pam_core/modules/sponge_layer.h:8-95), "next row" N2 of SURVEY.md section 8f.
 * fields: num_fields pointers to (nz,ny,nx,nens) arrays in the reference's order density_dry, uvel, vvel, wvel, temp,
 * tracers...; zint (nz+1,nens), zmid (nz,nens).  The horizontal means are accumulated in the reference's serial
 * atomicAdd order (j outer, i inner); field index 3 (wvel) relaxes towards zero (:34, :76). */
void awfl_oracle_sponge_layer(int nens, int nx, int ny, int nz, int num_fields, double *const *fields, const double *zint,
                              const double *zmid, double dt, int num_layers, double time_scale) {
  const int WFLD = 3;
  const double r_nx_ny = 1.0 / (nx * ny);
  double *havg = (double *)calloc((size_t)num_fields * nz * nens, sizeof(double));
  for (int ifld = 0; ifld < num_fields; ifld++)
    for (int kloc = 0; kloc < num_layers; kloc++)
      for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) for (int e = 0; e < nens; e++) {
        int k = nz - 1 - kloc;
        if (ifld != WFLD) havg[((size_t)ifld * nz + k) * nens + e] += fields[ifld][C4(k, j, i, e)] * r_nx_ny;
      }
  const double time_factor = dt / time_scale;
  for (int ifld = 0; ifld < num_fields; ifld++)
    for (int kloc = 0; kloc < num_layers; kloc++)
      for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) for (int e = 0; e < nens; e++) {
        int k = nz - 1 - kloc;
        double rel_dist = (zint[(size_t)nz * nens + e] - zmid[(size_t)k * nens + e]) /
                          (zint[(size_t)nz * nens + e] - zmid[(size_t)(nz - 1 - (num_layers - 1)) * nens + e]);
        double space_factor = (cos(M_PI * rel_dist) + 1) / 2;
        double factor = space_factor * time_factor;
        double *f = fields[ifld];
        f[C4(k, j, i, e)] += (havg[((size_t)ifld * nz + k) * nens + e] - f[C4(k, j, i, e)]) * factor;
      }
  free(havg);
}

/* ---------------------------------------------------------------------------------------------- */
/* Optional idealised initial conditions of Dycore::init (Dycore.h:986-1090: data_spec thermal / supercell).
 * Both fill the halo'd dycore state and finish with convert_dynamics_to_coupler (Dycore.h:1089). */
static double sample_ellipse_cosine(double amp, double x, double y, double z, double x0, double y0, double z0,
                                    double xrad, double yrad, double zrad) { /* Dycore.h:753-766 */
  double dist = sqrt(((x - x0) / xrad) * ((x - x0) / xrad) + ((y - y0) / yrad) * ((y - y0) / yrad) +
                     ((z - z0) / zrad) * ((z - z0) / zrad)) * M_PI / 2.;
  if (dist <= M_PI / 2.) return amp * pow(cos(dist), 2.0);
  return 0.;
}
static void hydro_const_theta(double z, double grav, double C0, double cp, double p0, double gamma, double rd,
                              double *r, double *t) { /* Dycore.h:739-748 */
  const double theta0 = 300., exner0 = 1.;
  *t = theta0;
  double exner = exner0 - grav * z / (cp * theta0);
  double p = p0 * pow(exner, (cp / rd));
  double rt = pow((p / C0), (1.0 / gamma));
  *r = rt / *t;
}

/* Dycore.h:1021-1088 (DATA_SPEC_THERMAL).  zmid (nz,nens). */
void awfl_oracle_init_thermal(const awfl_oracle_t *o, const double *zmid, double *rho_d_c, double *u_c, double *v_c,
                              double *w_c, double *temp_c, double *tracers_c) {
  const int nens = o->nens, nx = o->nx, ny = o->ny, nz = o->nz, nt = o->nt;
  const double dx = o->xlen / nx, dy = o->ylen / ny, xlen = o->xlen, ylen = o->ylen;
  const int sim2d = (ny == 1);
  const double qpoints[9] = AWFL_GLL9_PTS_INIT, qweights[9] = AWFL_GLL9_WTS_INIT;
  size_t nh = awfl_oracle_halo_elems(o);
  double *state = alloc_nan(NUM_STATE * nh), *tracers = alloc_nan((size_t)nt * nh);
  double *hyd = alloc_nan((size_t)nz * nens), *hyp = alloc_nan((size_t)nz * nens);
  for (int k = 0; k < nz; k++) for (int e = 0; e < nens; e++) {
    hyd[(size_t)k * nens + e] = 0.; hyp[(size_t)k * nens + e] = 0.;
    for (int kk = 0; kk < 9; kk++) {
      double z = zmid[(size_t)k * nens + e] + qpoints[kk] * DZ(k, e), hr, ht;
      hydro_const_theta(z, o->grav, o->C0, o->cp_d, o->p0, o->gamma_d, o->R_d, &hr, &ht);
      hyd[(size_t)k * nens + e] += hr * qweights[kk];
      hyp[(size_t)k * nens + e] += o->C0 * pow(hr * ht, o->gamma_d) * qweights[kk];
    }
  }
#pragma omp parallel for collapse(2) schedule(static)
  for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) for (int e = 0; e < nens; e++) {
    for (int l = 0; l < NUM_STATE; l++) state[H5(l, HS + k, HS + j, HS + i, e)] = 0.;
    for (int l = 0; l < nt; l++) tracers[H5(l, HS + k, HS + j, HS + i, e)] = 0.;
    for (int kk = 0; kk < 9; kk++) for (int jj = 0; jj < 9; jj++) for (int ii = 0; ii < 9; ii++) {
      double x = (i + 0.5) * dx + qpoints[ii] * dx;
      double y = (j + 0.5) * dy + qpoints[jj] * dy; if (sim2d) y = ylen / 2;
      double z = zmid[(size_t)k * nens + e] + qpoints[kk] * DZ(k, e);
      double hr = hyd[(size_t)k * nens + e], hp = hyp[(size_t)k * nens + e];
      double ht = pow(hp / o->C0, 1.0 / o->gamma_d) / hr;
      double rho = hr, u = 0, v = 0, w = 0, rho_v = 0;
      double theta = ht + sample_ellipse_cosine(2.0, x, y, z, xlen / 2, ylen / 2, 2000., 2000., 2000., 2000.);
      if (sim2d) v = 0;
      double wt = qweights[ii] * qweights[jj] * qweights[kk];
      state[H5(ID_R, HS + k, HS + j, HS + i, e)] += rho * wt;
      state[H5(ID_U, HS + k, HS + j, HS + i, e)] += rho * u * wt;
      state[H5(ID_V, HS + k, HS + j, HS + i, e)] += rho * v * wt;
      state[H5(ID_W, HS + k, HS + j, HS + i, e)] += rho * w * wt;
      state[H5(ID_T, HS + k, HS + j, HS + i, e)] += rho * theta * wt;
      for (int tr = 0; tr < nt; tr++) {
        if (tr == o->idWV) tracers[H5(tr, HS + k, HS + j, HS + i, e)] += rho_v * wt;
        else tracers[H5(tr, HS + k, HS + j, HS + i, e)] += 0 * wt;
      }
    }
  }
  awfl_oracle_convert_dynamics_to_coupler(o, state, tracers, rho_d_c, u_c, v_c, w_c, temp_c, tracers_c);
  free(state); free(tracers); free(hyd); free(hyp);
}

/* Dycore.h:777-830 supercell profile helpers */
static double sc_temperature(double z, double z_0, double z_trop, double z_top, double T_0, double T_trop, double T_top) {
  if (z <= z_trop) { double lapse = -(T_trop - T_0) / (z_trop - z_0); return T_0 - lapse * (z - z_0); }
  double lapse = -(T_top - T_trop) / (z_top - z_trop);
  return T_trop - lapse * (z - z_trop);
}
static double sc_pressure_dry(double z, double z_0, double z_trop, double z_top, double T_0, double T_trop, double T_top,
                              double p_0, double R_d, double grav) {
  if (z <= z_trop) {
    double lapse = -(T_trop - T_0) / (z_trop - z_0);
    double T = sc_temperature(z, z_0, z_trop, z_top, T_0, T_trop, T_top);
    return p_0 * pow(T / T_0, grav / (R_d * lapse));
  }
  double lapse = -(T_trop - T_0) / (z_trop - z_0);
  double p_trop = p_0 * pow(T_trop / T_0, grav / (R_d * lapse));
  lapse = -(T_top - T_trop) / (z_top - z_trop);
  if (lapse != 0) {
    double T = sc_temperature(z, z_0, z_trop, z_top, T_0, T_trop, T_top);
    return p_trop * pow(T / T_trop, grav / (R_d * lapse));
  }
  return p_trop * exp(-grav * (z - z_trop) / (R_d * T_trop));
}
static double sc_relhum(double z, double z_0, double z_trop) {
  if (z <= z_trop) return 1.0 - 0.75 * pow(z / z_trop, 1.25);
  return 0.25;
}
static double sc_sat_mix_dry(double press, double T) { return 380 / (press)*exp(17.27 * (T - 273) / (T - 36)); }

/* Dycore.h:1096-1276 init_supercell.  zmid (nz,nens), zint (nz+1,nens). */
void awfl_oracle_init_supercell(const awfl_oracle_t *o, const double *zmid, const double *zint, double *rho_d_c,
                                double *u_c, double *v_c, double *w_c, double *temp_c, double *tracers_c) {
  const int nens = o->nens, nx = o->nx, ny = o->ny, nz = o->nz, nt = o->nt;
  const double dx = o->xlen / nx, dy = o->ylen / ny, ylen = o->ylen;
  const int sim2d = (ny == 1);
  const double z_0 = 0, z_trop = 12000, T_0 = 300, T_trop = 213, T_top = 213, p_0 = 100000;
  const double gll_pts[9] = AWFL_GLL9_PTS_INIT, gll_wts[9] = AWFL_GLL9_WTS_INIT;
  const double R_d = o->R_d, R_v = o->R_v, grav = o->grav, gamma = o->gamma_d, C0 = o->C0;
  size_t nh = awfl_oracle_halo_elems(o);
  double *state = alloc_nan(NUM_STATE * nh), *tracers = alloc_nan((size_t)nt * nh);
  double *hyd = alloc_nan((size_t)nz * nens), *hyp = alloc_nan((size_t)nz * nens);
  double *quad = alloc_nan((size_t)nz * 8 * 9 * nens), *pg = alloc_nan((size_t)nz * 9 * nens);
  double *dg = alloc_nan((size_t)nz * 9 * nens), *dtg = alloc_nan((size_t)nz * 9 * nens), *dvg = alloc_nan((size_t)nz * 9 * nens);
#define QT(k, kk, kkk, e) quad[((((size_t)(k)) * 8 + (kk)) * 9 + (kkk)) * nens + (e)]
#define G3(a, k, kk, e) a[(((size_t)(k)) * 9 + (kk)) * nens + (e)]
#define ZTOP(e) zint[(size_t)nz * nens + (e)]
  for (int k = 0; k < nz; k++) for (int kk = 0; kk < 8; kk++) for (int kkk = 0; kkk < 9; kkk++) for (int e = 0; e < nens; e++) {
    double cellmid = zmid[(size_t)k * nens + e];
    double ngll_b = cellmid + gll_pts[kk] * DZ(k, e), ngll_t = cellmid + gll_pts[kk + 1] * DZ(k, e);
    double ngll_m = 0.5 * (ngll_b + ngll_t);
    double ngll_dz = DZ(k, e) * (gll_pts[kk + 1] - gll_pts[kk]);
    double zloc = ngll_m + ngll_dz * gll_pts[kkk];
    double temp = sc_temperature(zloc, z_0, z_trop, ZTOP(e), T_0, T_trop, T_top);
    double press_dry = sc_pressure_dry(zloc, z_0, z_trop, ZTOP(e), T_0, T_trop, T_top, p_0, R_d, grav);
    double qvs = sc_sat_mix_dry(press_dry, temp);
    double relhum = sc_relhum(zloc, z_0, z_trop);
    if (relhum * qvs > 0.014) relhum = 0.014 / qvs;
    double qv = fmin(0.014, qvs * relhum);
    QT(k, kk, kkk, e) = -(1 + qv) * grav / (R_d + qv * R_v) / temp;
  }
  for (int e = 0; e < nens; e++) {
    G3(pg, 0, 0, e) = p_0;
    for (int k = 0; k < nz; k++) for (int kk = 0; kk < 8; kk++) {
      double tot = 0;
      for (int kkk = 0; kkk < 9; kkk++) tot += QT(k, kk, kkk, e) * gll_wts[kkk];
      tot *= DZ(k, e) * (gll_pts[kk + 1] - gll_pts[kk]);
      G3(pg, k, kk + 1, e) = G3(pg, k, kk, e) * exp(tot);
      if (kk == 7 && k < nz - 1) G3(pg, k + 1, 0, e) = G3(pg, k, 8, e);
    }
  }
  for (int k = 0; k < nz; k++) for (int kk = 0; kk < 9; kk++) for (int e = 0; e < nens; e++) {
    double zloc = zmid[(size_t)k * nens + e] + gll_pts[kk] * DZ(k, e);
    double temp = sc_temperature(zloc, z_0, z_trop, ZTOP(e), T_0, T_trop, T_top);
    double press_tmp = sc_pressure_dry(zloc, z_0, z_trop, ZTOP(e), T_0, T_trop, T_top, p_0, R_d, grav);
    double qvs = sc_sat_mix_dry(press_tmp, temp);
    double relhum = sc_relhum(zloc, z_0, z_trop);
    if (relhum * qvs > 0.014) relhum = 0.014 / qvs;
    double qv = fmin(0.014, qvs * relhum);
    double press = G3(pg, k, kk, e);
    double dens_dry = press / (R_d + qv * R_v) / temp, dens_vap = qv * dens_dry;
    G3(dg, k, kk, e) = dens_dry + dens_vap;
    G3(dtg, k, kk, e) = pow(press / C0, 1.0 / gamma);
    G3(dvg, k, kk, e) = dens_vap;
  }
  /* cell means; the reference's inner `for iens` broadcast (:1226-1229) writes the same per-(k,iens) value for every
   * member because the lambda index shadows it -- the LAST outer iens wins for all members (quirk Q10). */
  for (int k = 0; k < nz; k++) for (int e = 0; e < nens; e++) {
    double press_tot = 0, dens_tot = 0;
    for (int kk = 0; kk < 9; kk++) { press_tot += G3(pg, k, kk, e) * gll_wts[kk]; dens_tot += G3(dg, k, kk, e) * gll_wts[kk]; }
    for (int e2 = 0; e2 < nens; e2++) { hyd[(size_t)k * nens + e2] = dens_tot; hyp[(size_t)k * nens + e2] = press_tot; }
  }
  for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) for (int e = 0; e < nens; e++) {
    for (int l = 0; l < NUM_STATE; l++) state[H5(l, HS + k, HS + j, HS + i, e)] = 0;
    for (int tr = 0; tr < nt; tr++) tracers[H5(tr, HS + k, HS + j, HS + i, e)] = 0;
    double pres = hyp[(size_t)k * nens + e];
    state[H5(ID_R, HS + k, HS + j, HS + i, e)] = hyd[(size_t)k * nens + e];
    state[H5(ID_T, HS + k, HS + j, HS + i, e)] = pow(pres / C0, 1.0 / gamma);
    for (int kk = 0; kk < 9; kk++) for (int jj = 0; jj < 9; jj++) for (int ii = 0; ii < 9; ii++) {
      double zloc = zmid[(size_t)k * nens + e] + gll_pts[kk] * DZ(k, e);
      const double zs = 5000, us = 30, uc = 15;
      double uvel = zloc < zs ? us * (zloc / zs) - uc : us - uc;
      double vvel = 0, wvel = 0;
      double dens_vap = G3(dvg, k, kk, e);
      double factor = gll_wts[ii] * gll_wts[jj] * gll_wts[kk];
      double r = state[H5(ID_R, HS + k, HS + j, HS + i, e)];
      state[H5(ID_U, HS + k, HS + j, HS + i, e)] += r * uvel * factor;
      state[H5(ID_V, HS + k, HS + j, HS + i, e)] += r * vvel * factor;
      state[H5(ID_W, HS + k, HS + j, HS + i, e)] += r * wvel * factor;
      tracers[H5(o->idWV, HS + k, HS + j, HS + i, e)] += dens_vap * factor;
    }
  }
  (void)dx; (void)dy; (void)ylen; (void)sim2d;
  awfl_oracle_convert_dynamics_to_coupler(o, state, tracers, rho_d_c, u_c, v_c, w_c, temp_c, tracers_c);
  free(state); free(tracers); free(hyd); free(hyp); free(quad); free(pg); free(dg); free(dtg); free(dvg);
#undef QT
#undef G3
#undef ZTOP
}

/* ---------------------------------------------------------------------------------------------- */
/* standalone/mmf_simplified/supercell_init.h:7-135: the driver's supercell column (hydrostatically integrated total
 * pressure on 5 Gauss-Lobatto points per cell, then 5-point cell averages of dry density, winds, temperature and vapour
 * density).  Host code; zint: nz+1 interface heights; outputs: nz values each. */
void awfl_oracle_supercell_init(int nz, const double *zint, double Rd, double Rv, double grav, double *rho_d_col,
                                double *uvel_col, double *vvel_col, double *wvel_col, double *temp_col, double *rho_v_col) {
  enum { ord = 5 };
  const double gll_pts[ord] = {-0.50000000000000000000000000000000000000, -0.32732683535398857189914622812342917778,
                               0.00000000000000000000000000000000000000, 0.32732683535398857189914622812342917778,
                               0.50000000000000000000000000000000000000};
  const double gll_wts[ord] = {0.050000000000000000000000000000000000000, 0.27222222222222222222222222222222222222,
                               0.35555555555555555555555555555555555556, 0.27222222222222222222222222222222222222,
                               0.050000000000000000000000000000000000000};
  const double z_0 = 0, z_trop = 12000, T_0 = 300, T_trop = 213, T_top = 213, p_0 = 100000;
  double *quad_temp = (double *)malloc((size_t)nz * (ord - 1) * ord * sizeof(double));
  double *hyp = (double *)malloc((size_t)nz * ord * sizeof(double));
  const double ztop = zint[nz];
  for (int k = 0; k < nz; k++)                                             /* :46-66 */
    for (int kk = 0; kk < ord - 1; kk++)
      for (int kkk = 0; kkk < ord; kkk++) {
        double dz = zint[k + 1] - zint[k];
        double cellmid = zint[k] + 0.5 * dz;
        double ord_b = cellmid + gll_pts[kk] * dz;
        double ord_t = cellmid + gll_pts[kk + 1] * dz;
        double ord_m = 0.5 * (ord_b + ord_t);
        double ord_dz = dz * (gll_pts[kk + 1] - gll_pts[kk]);
        double zloc = ord_m + ord_dz * gll_pts[kkk];
        double temp = sc_temperature(zloc, z_0, z_trop, ztop, T_0, T_trop, T_top);
        double press_dry = sc_pressure_dry(zloc, z_0, z_trop, ztop, T_0, T_trop, T_top, p_0, Rd, grav);
        double qvs = sc_sat_mix_dry(press_dry, temp);
        double relhum = sc_relhum(zloc, z_0, z_trop);
        if (relhum * qvs > 0.014) relhum = 0.014 / qvs;
        double qv = fmin(0.014, qvs * relhum);                             /* std::min(0.014, NaN) == 0.014 */
        quad_temp[((size_t)k * (ord - 1) + kk) * ord + kkk] = -(1 + qv) * grav / (Rd + qv * Rv) / temp;
      }
  hyp[0] = p_0;                                                            /* :70-87 */
  for (int k = 0; k < nz; k++) {
    double dz = zint[k + 1] - zint[k];
    for (int kk = 0; kk < ord - 1; kk++) {
      double tot = 0;
      for (int kkk = 0; kkk < ord; kkk++) tot += quad_temp[((size_t)k * (ord - 1) + kk) * ord + kkk] * gll_wts[kkk];
      tot *= dz * (gll_pts[kk + 1] - gll_pts[kk]);
      hyp[(size_t)k * ord + kk + 1] = hyp[(size_t)k * ord + kk] * exp(tot);
      if (kk == ord - 2 && k < nz - 1) hyp[(size_t)(k + 1) * ord] = hyp[(size_t)k * ord + ord - 1];
    }
  }
  for (int k = 0; k < nz; k++) {                                           /* :92-133 */
    rho_d_col[k] = 0; uvel_col[k] = 0; vvel_col[k] = 0; wvel_col[k] = 0; temp_col[k] = 0; rho_v_col[k] = 0;
    for (int kk = 0; kk < ord; kk++) {
      double dz = zint[k + 1] - zint[k];
      double zmid = 0.5 * (zint[k] + zint[k + 1]);
      double zloc = zmid + gll_pts[kk] * dz;
      double temp = sc_temperature(zloc, z_0, z_trop, ztop, T_0, T_trop, T_top);
      double press_dry = sc_pressure_dry(zloc, z_0, z_trop, ztop, T_0, T_trop, T_top, p_0, Rd, grav);
      double qvs = sc_sat_mix_dry(press_dry, temp);
      double relhum = sc_relhum(zloc, z_0, z_trop);
      if (relhum * qvs > 0.014) relhum = 0.014 / qvs;
      double qv = fmin(0.014, qvs * relhum);
      double p = hyp[(size_t)k * ord + kk];
      double rho_d = p / (Rd + qv * Rv) / temp;
      double rho_v = qv * rho_d;
      double uvel;
      const double zs = 5000, us = 30, uc = 15;
      if (zloc < zs) uvel = us * (zloc / zs) - uc;
      else uvel = us - uc;
      double vvel = 0, wvel = 0;
      rho_d_col[k] += rho_d * gll_wts[kk];
      uvel_col[k] += uvel * gll_wts[kk];
      vvel_col[k] += vvel * gll_wts[kk];
      wvel_col[k] += wvel * gll_wts[kk];
      temp_col[k] += temp * gll_wts[kk];
      rho_v_col[k] += rho_v * gll_wts[kk];
    }
  }
  free(quad_temp); free(hyp);
}

/* Kessler microphysics, "next row" N4 (physics/micro/kessler/Microphysics.h:120-268 timeStep, :346-457 kessler()).
 * Arrays are the coupler's (nz, ncol) collapsed views, ncol = ny*nx*nens (get_lev_col); zmid (nz,nens); precl (ncol).
 * rainsplit_in > 0 overrides the sub-cycle count (ensemble shards must agree on the global minimum).  Returns rainsplit. */
int awfl_oracle_kessler(int nens, int nx, int ny, int nz, double *rho_v, double *rho_c, double *rho_r, const double *rho_dry,
                        double *temp, double *precl, const double *zmid_in, double dt, double R_d, double R_v, double cp_d,
                        double p0, int rainsplit_in) {
  const size_t ncol = (size_t)ny * nx * nens;
  const size_t n = (size_t)nz * ncol;
#define A2(a, k, i) a[(size_t)(k) * ncol + (i)]
  double *qv = alloc_nan(n), *qc = alloc_nan(n), *qr = alloc_nan(n), *pressure = alloc_nan(n), *theta = alloc_nan(n);
  double *exner = alloc_nan(n), *zmid = alloc_nan(n);
  for (int k = 0; k < nz; k++) for (size_t i = 0; i < ncol; i++) A2(zmid, k, i) = zmid_in[(size_t)k * nens + (i % nens)]; /* :156-158 */
  for (int k = 0; k < nz; k++) for (size_t i = 0; i < ncol; i++) {                                                        /* :167-174 */
    A2(qv, k, i) = A2(rho_v, k, i) / A2(rho_dry, k, i);
    A2(qc, k, i) = A2(rho_c, k, i) / A2(rho_dry, k, i);
    A2(qr, k, i) = A2(rho_r, k, i) / A2(rho_dry, k, i);
    A2(pressure, k, i) = R_d * A2(rho_dry, k, i) * A2(temp, k, i) + R_v * A2(rho_v, k, i) * A2(temp, k, i);
    A2(exner, k, i) = pow(A2(pressure, k, i) / p0, R_d / cp_d);
    A2(theta, k, i) = A2(temp, k, i) / A2(exner, k, i);
  }
  /* kessler(theta, qv, qc, qr, rho_dry, precl, zmid, exner, dt, R_d, cp_d, p0)  :346-457 */
  const double *rho = rho_dry, *z = zmid, *pk = exner;
  const double Rd = R_d, cp = cp_d;
  const double psl = p0 / 100, rhoqr = 1000., lv = 2.5e6;
  double *r = alloc_nan(n), *rhalf = alloc_nan(n), *pc = alloc_nan(n), *velqr = alloc_nan(n), *sed = alloc_nan(n);
  double dt_max = INFINITY;
  for (int k = 0; k < nz; k++) for (size_t i = 0; i < ncol; i++) {
    A2(r, k, i) = 0.001 * A2(rho, k, i);
    A2(rhalf, k, i) = sqrt(A2(rho, 0, i) / A2(rho, k, i));
    A2(pc, k, i) = 3.8 / (pow(A2(pk, k, i), cp / Rd) * psl);
    A2(velqr, k, i) = 36.34 * pow(A2(qr, k, i) * A2(r, k, i), 0.1364) * A2(rhalf, k, i);
    if (k == 0) precl[i] = 0;
  }
  for (int k = 0; k < nz - 1; k++) for (size_t i = 0; i < ncol; i++) {
    double d = (A2(velqr, k, i) > 1.e-10) ? 0.8 * (A2(z, k + 1, i) - A2(z, k, i)) / A2(velqr, k, i) : dt;
    if (d < dt_max) dt_max = d;
  }
  int rainsplit = rainsplit_in > 0 ? rainsplit_in : (int)ceil(dt / dt_max);
  double dt0 = dt / (double)rainsplit;
  for (int nt = 0; nt < rainsplit; nt++) {
    for (int k = 0; k < nz; k++) for (size_t i = 0; i < ncol; i++) {
      if (k == 0) precl[i] = precl[i] + A2(rho, 0, i) * A2(qr, 0, i) * A2(velqr, 0, i) / rhoqr;
      if (k == nz - 1) {
        A2(sed, nz - 1, i) = -dt0 * A2(qr, nz - 1, i) * A2(velqr, nz - 1, i) / (0.5 * (A2(z, nz - 1, i) - A2(z, nz - 2, i)));
      } else {
        A2(sed, k, i) = dt0 * (A2(r, k + 1, i) * A2(qr, k + 1, i) * A2(velqr, k + 1, i) - A2(r, k, i) * A2(qr, k, i) * A2(velqr, k, i)) /
                        (A2(r, k, i) * (A2(z, k + 1, i) - A2(z, k, i)));
      }
    }
    for (int k = 0; k < nz; k++) for (size_t i = 0; i < ncol; i++) {
      double qrprod = A2(qc, k, i) - (A2(qc, k, i) - dt0 * fmax(0.001 * (A2(qc, k, i) - 0.001), 0.)) /
                                         (1 + dt0 * 2.2 * pow(A2(qr, k, i), 0.875));
      A2(qc, k, i) = fmax(A2(qc, k, i) - qrprod, 0.);
      A2(qr, k, i) = fmax(A2(qr, k, i) + qrprod + A2(sed, k, i), 0.);
      double tmp = A2(pk, k, i) * A2(theta, k, i) - 36.;
      double qvs = A2(pc, k, i) * exp(17.27 * (A2(pk, k, i) * A2(theta, k, i) - 273.) / tmp);
      double prod = (A2(qv, k, i) - qvs) / (1. + qvs * (4093. * lv / cp) / (tmp * tmp));
      double tmp1 = dt0 * (((1.6 + 124.9 * pow(A2(r, k, i) * A2(qr, k, i), 0.2046)) * pow(A2(r, k, i) * A2(qr, k, i), 0.525)) /
                           (2550000. * A2(pc, k, i) / (3.8 * qvs) + 540000.)) *
                    (fmax(qvs - A2(qv, k, i), 0.) / (A2(r, k, i) * qvs));
      double tmp2 = fmax(-prod - A2(qc, k, i), 0.);
      double tmp3 = A2(qr, k, i);
      double ern = fmin(tmp1, fmin(tmp2, tmp3));
      A2(theta, k, i) = A2(theta, k, i) + lv / (cp * A2(pk, k, i)) * (fmax(prod, -A2(qc, k, i)) - ern);
      A2(qv, k, i) = fmax(A2(qv, k, i) - fmax(prod, -A2(qc, k, i)) + ern, 0.);
      A2(qc, k, i) = A2(qc, k, i) + fmax(prod, -A2(qc, k, i));
      A2(qr, k, i) = A2(qr, k, i) - ern;
      A2(velqr, k, i) = 36.34 * pow(A2(qr, k, i) * A2(r, k, i), 0.1364) * A2(rhalf, k, i);
      if (k == 0 && nt == rainsplit - 1) precl[i] = precl[i] / (double)rainsplit;
    }
  }
  for (int k = 0; k < nz; k++) for (size_t i = 0; i < ncol; i++) {   /* :243-250 */
    A2(rho_v, k, i) = A2(qv, k, i) * A2(rho_dry, k, i);
    A2(rho_c, k, i) = A2(qc, k, i) * A2(rho_dry, k, i);
    A2(rho_r, k, i) = A2(qr, k, i) * A2(rho_dry, k, i);
    A2(temp, k, i) = A2(theta, k, i) * A2(exner, k, i);
  }
  free(qv); free(qc); free(qr); free(pressure); free(theta); free(exner); free(zmid);
  free(r); free(rhalf); free(pc); free(velqr); free(sed);
#undef A2
  return rainsplit;
}

/* ---------------------------------------------------------------------------------------------- */
/* GCM forcing of the CRM mean state, "next row" N1 (pam_core/modules/gcm_forcing.h).  Sums run in the reference's serial
 * atomicAdd order: for each (k, iens) over j (outer), i (inner).
 *   crm[10]   (nz,ny,nx,nens): density_dry, uvel, vvel, temp, water_vapor, cloud_water, ice, cloud_water_num, ice_num, rain_num
 *   gcm[10]   (nz,nens): gcm_density_dry, gcm_uvel, gcm_vvel, gcm_temp, gcm_water_vapor, gcm_cloud_water, gcm_cloud_ice,
 *             gcm_num_liq, gcm_num_ice, gcm_num_rain
 *   tend[14]  (nz,nens): gcm_forcing_tend_{rho_d,uvel,vvel,temp,qtot,qv,ql,qi,rho_v,rho_l,rho_i,nc,ni,nr} */
enum { GF_RHOD, GF_U, GF_V, GF_T, GF_RV, GF_RL, GF_RI, GF_NC, GF_NI, GF_NR };
enum { GT_RHOD, GT_U, GT_V, GT_T, GT_QTOT, GT_QV, GT_QL, GT_QI, GT_RV, GT_RL, GT_RI, GT_NC, GT_NI, GT_NR };
#define K2(k, e) ((size_t)(k) * nens + (e))

/* compute_gcm_forcing_tendencies  (gcm_forcing.h:17-210); writes tend[] except GT_RV, GT_RL, GT_RI */
void awfl_oracle_gcm_forcing_compute(int nens, int nx, int ny, int nz, const double *const *crm, const double *const *gcm,
                                     double *const *tend, double dt_gcm) {
  const size_t n2 = (size_t)nz * nens;
  double *ca[10];
  for (int f = 0; f < 10; f++) ca[f] = (double *)calloc(n2, sizeof(double));
  const double r_nx_ny = 1.0 / (nx * ny);
  for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) for (int e = 0; e < nens; e++) {
    const size_t c = C4(k, j, i, e);
    ca[GF_RHOD][K2(k, e)] += crm[GF_RHOD][c] * r_nx_ny;
    ca[GF_U][K2(k, e)] += crm[GF_U][c] * r_nx_ny;
    ca[GF_V][K2(k, e)] += crm[GF_V][c] * r_nx_ny;
    ca[GF_T][K2(k, e)] += crm[GF_T][c] * r_nx_ny;
    double tmp_qv = crm[GF_RV][c] / (crm[GF_RHOD][c] + crm[GF_RV][c]);
    double tmp_ql = crm[GF_RL][c] / (crm[GF_RHOD][c] + crm[GF_RV][c]);
    double tmp_qi = crm[GF_RI][c] / (crm[GF_RHOD][c] + crm[GF_RV][c]);
    ca[GF_RV][K2(k, e)] += tmp_qv * r_nx_ny;
    ca[GF_RL][K2(k, e)] += tmp_ql * r_nx_ny;
    ca[GF_RI][K2(k, e)] += tmp_qi * r_nx_ny;
    ca[GF_NC][K2(k, e)] += crm[GF_NC][c] * r_nx_ny;
    ca[GF_NI][K2(k, e)] += crm[GF_NI][c] * r_nx_ny;
    ca[GF_NR][K2(k, e)] += crm[GF_NR][c] * r_nx_ny;
  }
  const double r_dt_gcm = 1.0 / dt_gcm;
  for (int k = 0; k < nz; k++) for (int e = 0; e < nens; e++) {
    const size_t c = K2(k, e);
    tend[GT_RHOD][c] = (gcm[GF_RHOD][c] - ca[GF_RHOD][c]) * r_dt_gcm;
    tend[GT_U][c] = (gcm[GF_U][c] - ca[GF_U][c]) * r_dt_gcm;
    tend[GT_V][c] = (gcm[GF_V][c] - ca[GF_V][c]) * r_dt_gcm;
    tend[GT_T][c] = (gcm[GF_T][c] - ca[GF_T][c]) * r_dt_gcm;
    double tmp_qv_gcm = gcm[GF_RV][c] / (gcm[GF_RHOD][c] + gcm[GF_RV][c]);
    double tmp_ql_gcm = gcm[GF_RL][c] / (gcm[GF_RHOD][c] + gcm[GF_RV][c]);
    double tmp_qi_gcm = gcm[GF_RI][c] / (gcm[GF_RHOD][c] + gcm[GF_RV][c]);
    tend[GT_QV][c] = (tmp_qv_gcm - ca[GF_RV][c]) * r_dt_gcm;
    tend[GT_QL][c] = (tmp_ql_gcm - ca[GF_RL][c]) * r_dt_gcm;
    tend[GT_QI][c] = (tmp_qi_gcm - ca[GF_RI][c]) * r_dt_gcm;
    tend[GT_NC][c] = (gcm[GF_NC][c] - ca[GF_NC][c]) * r_dt_gcm;
    tend[GT_NI][c] = (gcm[GF_NI][c] - ca[GF_NI][c]) * r_dt_gcm;
    tend[GT_NR][c] = (gcm[GF_NR][c] - ca[GF_NR][c]) * r_dt_gcm;
    tend[GT_QTOT][c] = tend[GT_QV][c] + tend[GT_QL][c] + tend[GT_QI][c];
  }
  for (int f = 0; f < 10; f++) free(ca[f]);
}

static double yakl_max(double a, double b) { return a > b ? a : b; }   /* yakl::max: NaN in b propagates */

/* fill_holes  (gcm_forcing.h:213-284).  Returns 1 when the whole-CRM fallback ran. */
static int gcm_fill_holes(int nens, int nx, int ny, int nz, const double *dz, double *rho_x, const double *neg_mass) {
  const size_t n2 = (size_t)nz * nens;
  double *pos_mass = (double *)calloc(n2, sizeof(double));
  for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) for (int e = 0; e < nens; e++)
    if (rho_x[C4(k, j, i, e)] > 0) pos_mass[K2(k, e)] += rho_x[C4(k, j, i, e)] * dz[K2(k, e)];
  int neg_too_large = 0;
  for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) for (int e = 0; e < nens; e++) {
    if (i == 0 && j == 0) { if (neg_mass[K2(k, e)] > pos_mass[K2(k, e)]) neg_too_large = 1; }
    if (pos_mass[K2(k, e)] > 0) {
      double factor = rho_x[C4(k, j, i, e)] * dz[K2(k, e)] / pos_mass[K2(k, e)];
      rho_x[C4(k, j, i, e)] = yakl_max(0., rho_x[C4(k, j, i, e)] - (neg_mass[K2(k, e)] * factor) / dz[K2(k, e)]);
    }
  }
  if (neg_too_large) {
    double *neg_glob = (double *)calloc(nens, sizeof(double)), *pos_glob = (double *)calloc(nens, sizeof(double));
    for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) for (int e = 0; e < nens; e++) {
      if (i == 0 && j == 0) neg_glob[e] += yakl_max(0., neg_mass[K2(k, e)] - pos_mass[K2(k, e)]);
      pos_glob[e] += rho_x[C4(k, j, i, e)] * dz[K2(k, e)];
    }
    for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) for (int e = 0; e < nens; e++) {
      double factor = rho_x[C4(k, j, i, e)] * dz[K2(k, e)] / pos_glob[e];
      rho_x[C4(k, j, i, e)] = yakl_max(0., rho_x[C4(k, j, i, e)] - (neg_glob[e] * factor) / dz[K2(k, e)]);
    }
    free(neg_glob); free(pos_glob);
  }
  free(pos_mass);
  return neg_too_large;
}

/* apply_gcm_forcing_tendencies  (gcm_forcing.h:297-440).  crm[] updated in place; tend[GT_RV..GT_RI] written.
 * Returns a bit mask: bit s (0 vapour, 1 liquid, 2 ice) = hole filling ran, bit 4+s = its whole-CRM fallback ran. */
int awfl_oracle_gcm_forcing_apply(int nens, int nx, int ny, int nz, double *const *crm, const double *const *gcm,
                                  double *const *tend, const double *dz, double dt, double dt_gcm) {
  const size_t n2 = (size_t)nz * nens;
  double *neg[3], *colavg[3];
  for (int s = 0; s < 3; s++) { neg[s] = (double *)calloc(n2, sizeof(double)); colavg[s] = (double *)calloc(n2, sizeof(double)); }
  const double r_nx_ny = 1.0 / (nx * ny);
  for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) for (int e = 0; e < nens; e++) {
    const size_t c = C4(k, j, i, e), c2 = K2(k, e);
    double rho_d_old = crm[GF_RHOD][c];
    crm[GF_RHOD][c] += tend[GT_RHOD][c2] * dt;
    crm[GF_U][c] += tend[GT_U][c2] * dt;
    crm[GF_V][c] += tend[GT_V][c2] * dt;
    crm[GF_T][c] += tend[GT_T][c2] * dt;
    double tmp_qv_old = crm[GF_RV][c] / (rho_d_old + crm[GF_RV][c]);
    double tmp_ql_old = crm[GF_RL][c] / (rho_d_old + crm[GF_RV][c]);
    double tmp_qi_old = crm[GF_RI][c] / (rho_d_old + crm[GF_RV][c]);
    double tmp_qv_new = (tmp_qv_old + tend[GT_QV][c2] * dt);
    double tmp_ql_new = (tmp_ql_old + tend[GT_QL][c2] * dt);
    double tmp_qi_new = (tmp_qi_old + tend[GT_QI][c2] * dt);
    crm[GF_RV][c] = tmp_qv_new * crm[GF_RHOD][c] / (1 - tmp_qv_new);
    crm[GF_RL][c] = tmp_ql_new * (crm[GF_RHOD][c] + crm[GF_RV][c]);
    crm[GF_RI][c] = tmp_qi_new * (crm[GF_RHOD][c] + crm[GF_RV][c]);
    crm[GF_NC][c] += tend[GT_NC][c2] * dt;
    crm[GF_NI][c] += tend[GT_NI][c2] * dt;
    crm[GF_NR][c] += tend[GT_NR][c2] * dt;
    if (crm[GF_NC][c] < 0) crm[GF_NC][c] = 0;
    if (crm[GF_NI][c] < 0) crm[GF_NI][c] = 0;
    if (crm[GF_NR][c] < 0) crm[GF_NR][c] = 0;
    for (int s = 0; s < 3; s++) {
      colavg[s][c2] += crm[GF_RV + s][c] * r_nx_ny;
    }
    for (int s = 0; s < 3; s++)
      if (crm[GF_RV + s][c] < 0) {
        neg[s][c2] += -crm[GF_RV + s][c] * dz[c2];
        crm[GF_RV + s][c] = 0;
      }
  }
  const double r_dt_gcm = 1.0 / dt_gcm;
  for (size_t c2 = 0; c2 < n2; c2++)
    for (int s = 0; s < 3; s++) tend[GT_RV + s][c2] = (gcm[GF_RV + s][c2] - colavg[s][c2]) * r_dt_gcm;
  int mask = 0;
  for (int s = 0; s < 3; s++) {
    double sum = 0;
    for (size_t c2 = 0; c2 < n2; c2++) sum += neg[s][c2];
    if (sum > 0) {
      mask |= 1 << s;
      if (gcm_fill_holes(nens, nx, ny, nz, dz, crm[GF_RV + s], neg[s])) mask |= 16 << s;
    }
  }
  for (int s = 0; s < 3; s++) { free(neg[s]); free(colavg[s]); }
  return mask;
}

/* ---------------------------------------------------------------------------------------------- */
/* modules::broadcast_initial_gcm_column  (pam_core/modules/broadcast_initial_gcm_column.h:8-41; nfields = 6:
 * density_dry, uvel, vvel, wvel, temp, water_vapor <- gcm_density_dry, gcm_uvel, gcm_vvel, gcm_wvel, gcm_temp,
 * gcm_water_vapor) and ..._dry_density (:44-62; nfields = 1). */
void awfl_oracle_broadcast_gcm_column(int nens, int nx, int ny, int nz, int nfields, const double *const *gcm, double *const *crm) {
  for (int f = 0; f < nfields; f++)
    for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) for (int e = 0; e < nens; e++)
      crm[f][C4(k, j, i, e)] = gcm[f][K2(k, e)];
}

/* modules::perturb_temperature  (pam_core/modules/perturb_temperature.h:10-63).  The reference draws its numbers from
 * yakl::Random (third-party, absent: not reproducible); this restatement keeps everything else -- seed formula (:44),
 * [-1,1] range and clamp (:46-48), linear decay over the lowest nz/4 levels (:49), horizontal means summed in serial
 * order and the energy-conserving rescale (:33-36,:51,:55-60) -- and takes splitmix64(seed) as the generator. */
static double splitmix64_unit(uint64_t seed) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
void awfl_oracle_perturb_temperature(int nens, int nx, int ny, int nz, double *temp, const int *id, double magnitude) {
  const int num_levels = nz / 4;
  const size_t n2 = (size_t)nz * nens;
  double *hmean1 = (double *)calloc(n2, sizeof(double)), *hmean2 = (double *)calloc(n2, sizeof(double));
  const double r_nx_ny = 1.0 / (nx * ny);
  for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) for (int e = 0; e < nens; e++)
    hmean1[K2(k, e)] += temp[C4(k, j, i, e)] * r_nx_ny;
  for (int k = 0; k < num_levels; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) for (int e = 0; e < nens; e++) {
    int64_t seed = (int64_t)id[e] * num_levels * ny * nx + (int64_t)k * ny * nx + (int64_t)j * nx + i;
    double rnd = splitmix64_unit((uint64_t)seed) * 2. - 1.;
    rnd = fmin(rnd, 1.0);
    rnd = fmax(rnd, -1.0);
    double scaling = (num_levels - (double)k) / num_levels;
    temp[C4(k, j, i, e)] += rnd * magnitude * scaling;
    hmean2[K2(k, e)] += temp[C4(k, j, i, e)] * r_nx_ny;
  }
  for (int k = 0; k < num_levels; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) for (int e = 0; e < nens; e++)
    temp[C4(k, j, i, e)] = temp[C4(k, j, i, e)] * hmean1[K2(k, e)] / hmean2[K2(k, e)];
  free(hmean1); free(hmean2);
}
