// modules_kernels.hip -- gfx950 kernels of the coupler modules around the dycore (include/pam_amd_modules.h).
// sponge_layer: pam_core/modules/sponge_layer.h:8-95.  Both kernels are tiny and HBM-bound (top 5 of 60 levels).
#include <hip/hip_runtime.h>

#include <cmath>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pam_amd_awfl.h"
#include "../../include/pam_amd_modules.h"
#include "awfl_device.h"       // pow_pos_fast + its tables (the step's own x^y for positive bases)
#include "awfl_vertical.h"     // build_pow_tab
#include "supercell_sounding.h"

namespace {

constexpr int MAX_FIELDS = 55;   // 5 state fields + pam_const.h:24 max_fields tracers
struct FieldPtrs { double *p[MAX_FIELDS]; };

// Horizontal sums of the modules (sponge_layer, gcm_forcing): the reference accumulates them with atomicAdd, in no particular order.
// Here every sum is deterministic and does not depend on how many members the call holds (a CRM must be bit-reproducible between a
// 1-GPU run and a run sharded by members over N GPUs): the ny*nx cells of a level are dealt to MOD_NS = 16 SLOTS -- cell c belongs to
// slot c % 16 --, a thread sums the cells of ONE slot of ONE member in ascending order, and the 16 slot sums of a member are added in
// ascending order through LDS (slot_reduce).  A workgroup = (level, block of up to 64 members) x 16 slots: lanes are consecutive
// members, so every step of a walk is one coalesced row per field, and the 16 wavefronts of a workgroup keep 16 rows of every field in
// flight.  No scratch arrays: round 5's strip sums lived in stream-ordered allocations (hipMallocAsync / hipFreeAsync per call), which
// on this runtime made the C++ driver's CRM loop irreproducible from run to run (tools/ci_variants.sh: the sponge layer relaxing
// towards garbage means whenever the host synchronised between modules).
constexpr int MOD_NS = 16;     // sponge_layer
// gcm_forcing: ten fields per cell and three divisions.  Slots per member chosen by measurement at 1024 x 32x32x60 (round 6): the
// column averages (reads only) 8 slots, no unrolling: 0.89 ms (16 slots: 1.19; 4: 1.06); the apply pass (every field read and written)
// 16 slots, no unrolling: 2.15 ms (8 slots: 2.30-2.42; 4: 2.26)
constexpr int GCM_NS = 8;
constexpr int GCM_NS_APPLY = 16;
// v[0..NQ) of every thread -> the member's totals (all NS threads of a member get them); red: NQ * NS * blockDim.x doubles of LDS
template <int NQ, int NS = MOD_NS>
__device__ __forceinline__ void slot_reduce(double (&v)[NQ], double *red) {
  constexpr int MOD_NS = NS;
  // rows of 64 members whatever the workgroup's width: every LDS address is the lane's own + a compile-time offset (with the runtime
  // width as the stride the compiler formed all NQ x NS addresses in registers first: 178 registers in the averages kernel, 684 B of
  // scratch per lane in the 1024-lane apply kernel)
  const int m = threadIdx.x, slot = threadIdx.y;
#pragma unroll
  for (int q = 0; q < NQ; q++) red[(q * MOD_NS + slot) * 64 + m] = v[q];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    double a = 0.0;
    // (a rolled loop, four reads in flight: unrolled, the compiler requests all NQ x NS values at once -- 2 x NQ x NS registers -- and
    // spills them; the sum runs once per thread)
#pragma unroll 4
    for (int sl = 0; sl < MOD_NS; sl++) a += red[(q * MOD_NS + sl) * 64 + m];
    v[q] = a;
  }
  __syncthreads();
}

// modules::sponge_layer (pam_core/modules/sponge_layer.h:8-95), mean and relaxation in ONE kernel: grid (member blocks, layers, fields).
// A workgroup sums its level of its field (wvel, field 3, keeps a zero mean: :34,:75), then relaxes the same cells -- their second
// read comes out of the caches (the top five levels of every field: 0.25 GB at 1024 x 32x32x60).
__global__ void __launch_bounds__(1024) sponge_kernel(FieldPtrs F, int nens, int ncol, int nz, int num_layers,
                                                      const double *__restrict__ zint, const double *__restrict__ zmid,
                                                      double time_factor) {
  __shared__ double red[MOD_NS * 64];
  const int ME = blockDim.x, slot = threadIdx.y;
  const int e0 = (int)blockIdx.x * ME + (int)threadIdx.x;
  const bool ok = e0 < nens;
  const int e = ok ? e0 : nens - 1;
  const int kloc = (int)blockIdx.y, ifld = (int)blockIdx.z, k = nz - 1 - kloc;
  double *f = F.p[ifld] + (long long)k * ncol * nens + e;
  double h[1] = {0.0};
  if (ifld != 3) {
    const double r_nx_ny = 1.0 / ncol;
#pragma unroll 8
    for (int c = slot; c < ncol; c += MOD_NS) h[0] += f[(long long)c * nens] * r_nx_ny;
  }
  slot_reduce<1>(h, red);
  const double ztop = zint[(long long)nz * nens + e];
  const double rel_dist = (ztop - zmid[(long long)k * nens + e]) / (ztop - zmid[(long long)(nz - 1 - (num_layers - 1)) * nens + e]);
  const double space_factor = (cos(M_PI * rel_dist) + 1) / 2;
  const double factor = space_factor * time_factor;
  if (!ok) return;
#pragma unroll 8
  for (int c = slot; c < ncol; c += MOD_NS) {
    const double v = f[(long long)c * nens];
    f[(long long)c * nens] = v + (h[0] - v) * factor;
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Kessler microphysics (physics/micro/kessler/Microphysics.h:120-268 timeStep, :346-457 kessler()).
// Columns are independent; col = (j*nx+i)*nens+e is the fastest index of every (nz, ncol) array, so consecutive lanes
// read consecutive doubles at every level.

// Every x^y of the scheme has a non-negative base: it goes through pow_pos_fast (awfl_device.h: ~65 instructions, 0.52 ulp against
// 80-bit powl, 0 -> 0) instead of the device library's pow (~260-440 instructions, half of them for negative / special bases) -- six of
// them per cell and sub-cycle made the column kernel VALU-bound (round 5: 4.2 -> 2.9 ms per timeStep at 1024 x 32x32x60).  T: its
// tables, staged in LDS by the kernels (two dependent per-lane look-ups per pow).
using pama::PowTab;
__device__ __forceinline__ double kpow(double x, double y, const PowTab *T) { return pama::pow_pos_fast(x, y, T); }
__device__ __forceinline__ double krcp(double x) { return pama::fast_rcp(x); }
__device__ __forceinline__ double kdiv(double a, double b) { return a * pama::fast_rcp(b); }
__device__ __forceinline__ void kessler_stage_tab(const PowTab *__restrict__ src, PowTab *dst) {
  const double *s = reinterpret_cast<const double *>(src);
  double *d = reinterpret_cast<double *>(dst);
  for (int i = threadIdx.x; i < (int)(sizeof(PowTab) / sizeof(double)); i += blockDim.x) d[i] = s[i];
  __syncthreads();
}
// x^y (y > 0) where the base is an amount of rain: most cells of most columns hold none and 0^y = 0 exactly, so a wavefront without rain
// skips the evaluation (a branch over ~65 instructions: taken per wavefront); any other base -- negative and NaN included -- goes
// through kpow as before.  Same values either way.
__device__ __forceinline__ double kpow_rain(double x, double y, const PowTab *T) {
  double r = 0.0;
  if (x != 0.0) r = kpow(x, y, T);
  return r;
}
__device__ __forceinline__ double kessler_velqr(double qr, double r, double rhalf, const PowTab *T) {
  return 36.34 * kpow_rain(qr * r, 0.1364, T) * rhalf;   // :375, :449
}

// The sedimentation time-step limit of kessler "main 1" (:376-386, the input of the global minimum :389-390); touches nothing.
// The minimum: wavefront shuffle reduce, then an atomicMin ONLY when the wavefront's value undercuts what the slot already
// holds -- ~1e6 wavefronts hammering one L2 address with unconditional atomics cost 11 ms at 1024 x 32x32x60, ten times
// the kernel's HBM time; the plain load in front leaves a handful.
// A workgroup takes 256 columns and every gridDim.y-th level: the pow tables are staged once per workgroup, not once per 256 cells
// (round 6: 0.63 -> ms at 1024 x 32x32x60, where the staging moved as many bytes as the two fields read).
__global__ void __launch_bounds__(256) kessler_limit_kernel(int nz, long long ncol, int nens, const double *__restrict__ rho_r,
                                                            const double *__restrict__ rho_dry, const double *__restrict__ zmid,
                                                            double dt, unsigned long long *dt_max_bits,
                                                            const PowTab *__restrict__ tab) {
  __shared__ PowTab sh_tab;
  kessler_stage_tab(tab, &sh_tab);
  const PowTab *PT = &sh_tab;
  const long long col = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  // positive doubles order like their bit patterns; 0 = "this state is not usable" (NaN or negative fall speed), which
  // wins every minimum and fails the host's "limit must be positive" test
  unsigned long long bits = ~0ull;
  if (col < ncol) {
    const int e = (int)(col % nens);
    const double rho0 = rho_dry[col];
    for (int k = blockIdx.y; k < nz - 1; k += gridDim.y) {
      const long long idx = (long long)k * ncol + col;
      const double rho = rho_dry[idx];
      const double qr = rho_r[idx] / rho;
      const double velqr = kessler_velqr(qr, 0.001 * rho, sqrt(rho0 / rho), PT);
      double dt2d = dt;
      if (velqr > 1.e-10) dt2d = 0.8 * (zmid[(long long)(k + 1) * nens + e] - zmid[(long long)k * nens + e]) / velqr;
      const unsigned long long b = (velqr >= 0 && dt2d > 0) ? (unsigned long long)__double_as_longlong(dt2d) : 0ull;
      bits = b < bits ? b : bits;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(bits, off);
    bits = o < bits ? o : bits;
  }
  if ((threadIdx.x & 63) == 0 && bits != ~0ull &&
      bits < __hip_atomic_load(dt_max_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMin(dt_max_bits, bits);
}

// The whole of timeStep for one column: the conversions :167-174 (densities -> mixing ratios, T -> theta through the Exner
// function of the incoming state), kessler "main 2" + "main 3" (:394-453) for all sub-cycles, the conversions back :243-250.  One
// thread marches one column upwards: sed(k) needs the not-yet-adjusted values of levels k and k+1, which an upward march has at
// hand.  The FIRST sub-cycle reads the coupler's arrays as they came (rounds 4-5 converted them in place in a kernel of their own:
// 11 more array passes of the 26); the LAST writes densities and temperature.  SINGLE (one sub-cycle, the usual case): nothing else
// is stored.  Otherwise the mixing ratios and theta live IN PLACE in the coupler arrays between sub-cycles and the Exner function
// of the incoming state in `exner` (nz x ncol doubles of scratch).  velqr, r, rhalf, pc are pure functions of stored values and are
// recomputed (bitwise the same as the reference's stored temporaries).
// IDX: unsigned when a field is below 2^29 doubles (one register of offset for all six arrays on top of their scalar bases).  The usual
// instance (one sub-cycle, 32-bit offsets) is held at four wavefronts per SIMD: 16384 single-wavefront workgroups of the C2 grid are
// then exactly four rounds of the chip (1.41 -> 1.28 ms; 20 bytes of scratch outside the level loop); the others keep three, which they
// reach without scratch.
template <bool SINGLE, class IDX>
__global__ void __launch_bounds__(64, (SINGLE && sizeof(IDX) == 4) ? 4 : 3) kessler_column_kernel(int nz, long long ncol, int nens, double *qv_a, double *qc_a,
                                                            double *qr_a, const double *__restrict__ rho_dry, double *theta_a,
                                                            double *precl, const double *__restrict__ zmid, double *exner,
                                                            double dt, int rainsplit, double Rd, double Rv, double cp, double p0,
                                                            const PowTab *__restrict__ tab) {
  __shared__ PowTab sh_tab;
  kessler_stage_tab(tab, &sh_tab);
  const PowTab *PT = &sh_tab;
  const long long col_ = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (col_ >= ncol) return;
  const IDX col = (IDX)col_, nc = (IDX)ncol;
  const int e = (int)(col_ % nens);
  const double psl = p0 / 100, rhoqr = 1000., lv = 2.5e6, rp0 = 1 / p0;
  const double dt0 = dt / (double)rainsplit;   // (uniform: scalar-side IEEE divisions stay)
  const double rho0 = rho_dry[col];
  double pr = 0;                                                                  // timeStep :176 precl = 0
  for (int nt = 0; nt < (SINGLE ? 1 : rainsplit); nt++) {
    const bool first = SINGLE || nt == 0, last = SINGLE || nt == rainsplit - 1;
    // level-k values carried from the previous iteration's "k+1" loads
    double rho_k = rho0, z_k = zmid[e], qr_k = qr_a[col];
    if (first) qr_k = kdiv(qr_k, rho_k);                                               // :169
    double r_k = 0.001 * rho_k, rhalf_k = sqrt(kdiv(rho0, rho_k));
    double vel_k = kessler_velqr(qr_k, r_k, rhalf_k, PT);
    double z_km1 = 0;
    pr = pr + rho0 * qr_k * vel_k / rhoqr;   // (a constant divisor: the compiler's reciprocal)                                       // :397
    for (int k = 0; k < nz; k++) {
      const IDX idx = (IDX)k * nc + col;
      double sed, rho_n = 0, z_n = 0, qr_n = 0, r_n = 0, rhalf_n = 0, vel_n = 0;
      if (k == nz - 1) {
        sed = kdiv(-dt0 * qr_k * vel_k, 0.5 * (z_k - z_km1));                        // :400
      } else {
        rho_n = rho_dry[idx + nc]; z_n = zmid[(long long)(k + 1) * nens + e]; qr_n = qr_a[idx + nc];
        if (first) qr_n = kdiv(qr_n, rho_n);
        r_n = 0.001 * rho_n; rhalf_n = sqrt(kdiv(rho0, rho_n));
        vel_n = kessler_velqr(qr_n, r_n, rhalf_n, PT);
        sed = kdiv(dt0 * (r_n * qr_n * vel_n - r_k * qr_k * vel_k), r_k * (z_n - z_k));   // :403
      }
      double qc = qc_a[idx], qv = qv_a[idx], theta = theta_a[idx], qr = qr_k, pk, pnorm = 0;
      if (first) {                                                                // :167-174
        const double rv = qv, T = theta;
        const double pressure = Rd * rho_k * T + Rv * rv * T;
        pnorm = pressure * rp0;
        pk = kpow(pnorm, Rd / cp, PT);
        const double rrho = krcp(rho_k);
        qv = rv * rrho; qc = qc * rrho; theta = kdiv(T, pk);
        if (!SINGLE) exner[idx] = pk;
      } else {
        pk = exner[idx];
      }
      // :374 pc = 3.8 / (pk^(cp/Rd) psl).  pk^(cp/Rd) IS pressure / p0 up to the rounding of two pows (a few ulp): where the pressure
      // is at hand (a first sub-cycle) the pow is not taken
      const double pc = kdiv(3.8, (first ? pnorm : kpow(pk, cp / Rd, PT)) * psl);
      // autoconversion and accretion (:412-414)
      const double qrprod = qc - kdiv(qc - dt0 * fmax(0.001 * (qc - 0.001), 0.), 1 + dt0 * 2.2 * kpow_rain(qr, 0.875, PT));
      qc = fmax(qc - qrprod, 0.);
      qr = fmax(qr + qrprod + sed, 0.);
      // saturation vapour mixing ratio (:417-422)
      const double tmp = pk * theta - 36.;
      const double rtmp = krcp(tmp);
      const double qvs = pc * exp(17.27 * (pk * theta - 273.) * rtmp);
      const double prod = kdiv(qv - qvs, 1. + qvs * (4093. * lv / cp) * (rtmp * rtmp));
      // evaporation of rain (:425-430)
      const double rq = r_k * qr;
      const double rqvs = krcp(qvs);
      double rq_a = 0.0, rq_b = 0.0;                            // rq^0.2046, rq^0.525: one logarithm for the two; none without rain
      if (rq != 0.0) {
        const pama::PowLog2 lrq = pama::pow_log2_dd(rq, PT);
        rq_a = pama::pow_exp2_dd(rq, 0.2046, lrq, PT);
        rq_b = pama::pow_exp2_dd(rq, 0.525, lrq, PT);
      }
      const double tmp1 = dt0 * kdiv((1.6 + 124.9 * rq_a) * rq_b, 2550000. * pc * (rqvs * (1 / 3.8)) + 540000.) *
                          (fmax(qvs - qv, 0.) * kdiv(rqvs, r_k));
      const double tmp2 = fmax(-prod - qc, 0.);
      const double ern = fmin(tmp1, fmin(tmp2, qr));
      // saturation adjustment (:433-439)
      const double cond = fmax(prod, -qc);
      theta = theta + kdiv(lv, cp * pk) * (cond - ern);
      qv = fmax(qv - cond + ern, 0.);
      qc = qc + cond;
      qr = qr - ern;
      if (last) {   // timeStep :243-250 (temp from the OLD Exner function)
        qv_a[idx] = qv * rho_k; qc_a[idx] = qc * rho_k; qr_a[idx] = qr * rho_k; theta_a[idx] = theta * pk;
      } else {
        qv_a[idx] = qv; qc_a[idx] = qc; qr_a[idx] = qr; theta_a[idx] = theta;
      }
      z_km1 = z_k;
      rho_k = rho_n; z_k = z_n; qr_k = qr_n; r_k = r_n; rhalf_k = rhalf_n; vel_k = vel_n;
    }
  }
  precl[col] = pr / (double)rainsplit;                                            // :452
}

// ---------------------------------------------------------------------------------------------------------------
// GCM forcing of the CRM mean state (pam_core/modules/gcm_forcing.h).  The reference accumulates its horizontal means
// and hole-filling masses with atomicAdd, in no particular order; here every sum is deterministic (slots: slot_reduce above).
// Consecutive lanes are consecutive members: each step of a walk is one coalesced row per field.
struct Gcm10 { double *p[10]; };
struct Gcm14 { double *p[14]; };
enum { GF_RHOD, GF_U, GF_V, GF_T, GF_RV, GF_RL, GF_RI, GF_NC, GF_NI, GF_NR };
enum { GT_RHOD, GT_U, GT_V, GT_T, GT_QTOT, GT_QV, GT_QL, GT_QI, GT_RV, GT_RL, GT_RI, GT_NC, GT_NI, GT_NR };

__device__ __forceinline__ double yakl_max(double a, double b) { return a > b ? a : b; }   // NaN in b propagates, as yakl::max

// tendencies of one (level, member) pair from its column averages (gcm_forcing.h:176-208)
__device__ __forceinline__ void gcm_forcing_compute_finish(const Gcm10 &gcm, const Gcm14 &tend, const double (&ca)[10], long long t,
                                                           double r_dt_gcm) {
  tend.p[GT_RHOD][t] = (gcm.p[GF_RHOD][t] - ca[GF_RHOD]) * r_dt_gcm;
  tend.p[GT_U][t] = (gcm.p[GF_U][t] - ca[GF_U]) * r_dt_gcm;
  tend.p[GT_V][t] = (gcm.p[GF_V][t] - ca[GF_V]) * r_dt_gcm;
  tend.p[GT_T][t] = (gcm.p[GF_T][t] - ca[GF_T]) * r_dt_gcm;
  const double den = gcm.p[GF_RHOD][t] + gcm.p[GF_RV][t];
  const double tqv = (gcm.p[GF_RV][t] / den - ca[GF_RV]) * r_dt_gcm;
  const double tql = (gcm.p[GF_RL][t] / den - ca[GF_RL]) * r_dt_gcm;
  const double tqi = (gcm.p[GF_RI][t] / den - ca[GF_RI]) * r_dt_gcm;
  tend.p[GT_QV][t] = tqv; tend.p[GT_QL][t] = tql; tend.p[GT_QI][t] = tqi;
  tend.p[GT_NC][t] = (gcm.p[GF_NC][t] - ca[GF_NC]) * r_dt_gcm;
  tend.p[GT_NI][t] = (gcm.p[GF_NI][t] - ca[GF_NI]) * r_dt_gcm;
  tend.p[GT_NR][t] = (gcm.p[GF_NR][t] - ca[GF_NR]) * r_dt_gcm;
  tend.p[GT_QTOT][t] = tqv + tql + tqi;
}

// compute_gcm_forcing_tendencies (gcm_forcing.h:17-210): column averages; grid (member blocks, levels), block (members, GCM_NS slots)
// IDX: unsigned when a field is below 2^29 doubles (byte offsets fit 32 bits: one register of address for all ten fields on top of
// their scalar bases), long long otherwise
template <class IDX>
__global__ void __launch_bounds__(64 * GCM_NS) gcm_forcing_compute_kernel(int nens, int ncol, int nz, Gcm10 crm, Gcm10 gcm, Gcm14 tend,
                                                                   double r_dt_gcm) {
  __shared__ double red[5 * GCM_NS * 64];
  const int ME = blockDim.x, slot = threadIdx.y;
  const int e0 = (int)blockIdx.x * ME + (int)threadIdx.x;
  const bool ok = e0 < nens;
  const int e = ok ? e0 : nens - 1, k = (int)blockIdx.y;
  const long long t = (long long)k * nens + e;
  const double r_nx_ny = 1.0 / ncol;
  double ca[10];
#pragma unroll
  for (int f = 0; f < 10; f++) ca[f] = 0;
  const IDX base = (IDX)k * (IDX)ncol * (IDX)nens + (IDX)e;
#pragma clang loop unroll(disable)
  for (int c = slot; c < ncol; c += GCM_NS) {
    const IDX o = base + (IDX)c * (IDX)nens;
    const double rd = crm.p[GF_RHOD][o], rv = crm.p[GF_RV][o];
    ca[GF_RHOD] += rd * r_nx_ny;
    ca[GF_U] += crm.p[GF_U][o] * r_nx_ny;
    ca[GF_V] += crm.p[GF_V][o] * r_nx_ny;
    ca[GF_T] += crm.p[GF_T][o] * r_nx_ny;
    ca[GF_RV] += (rv / (rd + rv)) * r_nx_ny;
    ca[GF_RL] += (crm.p[GF_RL][o] / (rd + rv)) * r_nx_ny;
    ca[GF_RI] += (crm.p[GF_RI][o] / (rd + rv)) * r_nx_ny;
    ca[GF_NC] += crm.p[GF_NC][o] * r_nx_ny;
    ca[GF_NI] += crm.p[GF_NI][o] * r_nx_ny;
    ca[GF_NR] += crm.p[GF_NR][o] * r_nx_ny;
  }
  double lo[5] = {ca[0], ca[1], ca[2], ca[3], ca[4]}, hi[5] = {ca[5], ca[6], ca[7], ca[8], ca[9]};
  slot_reduce<5, GCM_NS>(lo, red);
  slot_reduce<5, GCM_NS>(hi, red);
  if (slot == 0 && ok) {
    const double tot[10] = {lo[0], lo[1], lo[2], lo[3], lo[4], hi[0], hi[1], hi[2], hi[3], hi[4]};
    gcm_forcing_compute_finish(gcm, tend, tot, t, r_dt_gcm);
  }
}

// apply_gcm_forcing_tendencies, main kernel + diagnostics (gcm_forcing.h:361-429) fused with the first two kernels of
// fill_holes (positive mass per level, "negative too large" test; :236-250).
//   work: neg[3], pos[3] (nz,nens) ; flags[0..2] = some negative mass for species s, flags[3..5] = negative > positive somewhere
__device__ __forceinline__ void gcm_forcing_apply_finish(const Gcm10 &gcm, const Gcm14 &tend, const double (&colavg)[3],
                                                         const double (&neg)[3], const double (&pos)[3], long long t, long long n2,
                                                         double r_dt_gcm, double *__restrict__ work, int *__restrict__ flags) {
#pragma unroll
  for (int s = 0; s < 3; s++) {
    tend.p[GT_RV + s][t] = (gcm.p[GF_RV + s][t] - colavg[s]) * r_dt_gcm;
    work[(long long)s * n2 + t] = neg[s];
    work[(long long)(3 + s) * n2 + t] = pos[s];
    if (neg[s] > 0) atomicOr(&flags[s], 1);
    if (neg[s] > pos[s]) atomicOr(&flags[3 + s], 1);
  }
}
// grid (member blocks, levels), block (members, GCM_NS_APPLY slots): every cell is read and written once, the nine sums of a
// (level, member) pair -- colavg[3], neg[3], pos[3] -- go through slot_reduce
template <class IDX>
__global__ void __launch_bounds__(64 * GCM_NS_APPLY) gcm_forcing_apply_kernel(int nens, int ncol, int nz, Gcm10 crm, Gcm10 gcm, Gcm14 tend,
                                                                 const double *__restrict__ dz, double dt, double r_dt_gcm,
                                                                 double *__restrict__ work, int *__restrict__ flags) {
  __shared__ double red[3 * GCM_NS_APPLY * 64];
  const int ME = blockDim.x, slot = threadIdx.y;
  const int e0 = (int)blockIdx.x * ME + (int)threadIdx.x;
  const bool ok = e0 < nens;
  const int e = ok ? e0 : nens - 1, k = (int)blockIdx.y;
  const long long t = (long long)k * nens + e;
  const long long n2 = (long long)nz * nens;
  const double r_nx_ny = 1.0 / ncol;
  const double dzk = dz[t];
  const double t_rd = tend.p[GT_RHOD][t] * dt, t_u = tend.p[GT_U][t] * dt, t_v = tend.p[GT_V][t] * dt, t_t = tend.p[GT_T][t] * dt;
  const double t_qv = tend.p[GT_QV][t] * dt, t_ql = tend.p[GT_QL][t] * dt, t_qi = tend.p[GT_QI][t] * dt;
  const double t_nc = tend.p[GT_NC][t] * dt, t_ni = tend.p[GT_NI][t] * dt, t_nr = tend.p[GT_NR][t] * dt;
  double colavg[3] = {0, 0, 0}, neg[3] = {0, 0, 0}, pos[3] = {0, 0, 0};
  const IDX base = (IDX)k * (IDX)ncol * (IDX)nens + (IDX)e;
  if (ok) {
#pragma clang loop unroll(disable)
    for (int c = slot; c < ncol; c += GCM_NS_APPLY) {
      const IDX o = base + (IDX)c * (IDX)nens;
      const double rho_d_old = crm.p[GF_RHOD][o];
      const double rho_d = rho_d_old + t_rd;
      crm.p[GF_RHOD][o] = rho_d;
      crm.p[GF_U][o] += t_u;
      crm.p[GF_V][o] += t_v;
      crm.p[GF_T][o] += t_t;
      const double rv_old = crm.p[GF_RV][o];
      const double qv_new = rv_old / (rho_d_old + rv_old) + t_qv;
      const double ql_new = crm.p[GF_RL][o] / (rho_d_old + rv_old) + t_ql;
      const double qi_new = crm.p[GF_RI][o] / (rho_d_old + rv_old) + t_qi;
      double w[3];
      w[0] = qv_new * rho_d / (1 - qv_new);
      w[1] = ql_new * (rho_d + w[0]);
      w[2] = qi_new * (rho_d + w[0]);
      double nc = crm.p[GF_NC][o] + t_nc, ni = crm.p[GF_NI][o] + t_ni, nr = crm.p[GF_NR][o] + t_nr;   // :388-393
      if (nc < 0) nc = 0;
      if (ni < 0) ni = 0;
      if (nr < 0) nr = 0;
      crm.p[GF_NC][o] = nc; crm.p[GF_NI][o] = ni; crm.p[GF_NR][o] = nr;
#pragma unroll
      for (int s = 0; s < 3; s++) {
        colavg[s] += w[s] * r_nx_ny;
        if (w[s] < 0) { neg[s] += -w[s] * dzk; w[s] = 0; }
        if (w[s] > 0) pos[s] += w[s] * dzk;
        crm.p[GF_RV + s][o] = w[s];
      }
    }
  }
  // (three quantities at a time: 3 x 16 LDS values in flight per lane fit the 128 registers of a 1024-lane workgroup)
  slot_reduce<3, GCM_NS_APPLY>(colavg, red);
  slot_reduce<3, GCM_NS_APPLY>(neg, red);
  slot_reduce<3, GCM_NS_APPLY>(pos, red);
  if (slot == 0 && ok) gcm_forcing_apply_finish(gcm, tend, colavg, neg, pos, t, n2, r_dt_gcm, work, flags);
}

// fill_holes, level pass (:243-250)
__global__ void __launch_bounds__(256) gcm_fill_level_kernel(int nens, long long per_level, long long ncell, double *__restrict__ rho_x,
                                                             const double *__restrict__ dz, const double *__restrict__ neg,
                                                             const double *__restrict__ pos) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= ncell) return;
  const long long t = (idx / per_level) * nens + idx % nens;
  const double p = pos[t];
  if (p > 0) {
    const double d = dz[t], r = rho_x[idx];
    const double factor = r * d / p;
    rho_x[idx] = yakl_max(0.0, r - (neg[t] * factor) / d);
  }
}

// fill_holes, whole-CRM fallback: per-member sums in serial order (:262-266), one thread per member
__global__ void __launch_bounds__(64) gcm_fill_glob_sum_kernel(int nens, int nx, int ny, int nz, const double *__restrict__ rho_x,
                                                               const double *__restrict__ dz, const double *__restrict__ neg,
                                                               const double *__restrict__ pos, double *__restrict__ glob) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nens) return;
  double ng = 0, pg = 0;
  for (int k = 0; k < nz; k++) {
    const long long t = (long long)k * nens + e;
    ng += yakl_max(0.0, neg[t] - pos[t]);
    const double d = dz[t];
    const long long base = (long long)k * ny * nx * nens + e;
    for (int c = 0; c < ny * nx; c++) pg += rho_x[base + (long long)c * nens] * d;
  }
  glob[e] = ng;
  glob[nens + e] = pg;
}

// fill_holes, whole-CRM fallback: removal (:269-272)
__global__ void __launch_bounds__(256) gcm_fill_glob_kernel(int nens, long long per_level, long long ncell, double *__restrict__ rho_x,
                                                            const double *__restrict__ dz, const double *__restrict__ glob) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= ncell) return;
  const int e = (int)(idx % nens);
  const double d = dz[(idx / per_level) * nens + e], r = rho_x[idx];
  const double factor = r * d / glob[nens + e];
  rho_x[idx] = yakl_max(0.0, r - (glob[e] * factor) / d);
}

// ---------------------------------------------------------------------------------------------------------------
// modules::broadcast_initial_gcm_column[_dry_density]  (pam_core/modules/broadcast_initial_gcm_column.h:8-62)
struct Ptr6 { double *crm[6]; const double *gcm[6]; };
__global__ void __launch_bounds__(256) broadcast_gcm_kernel(int nens, long long per_level, long long ncell, int nfields, Ptr6 F) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= ncell) return;
  const long long t = (idx / per_level) * nens + idx % nens;
  for (int f = 0; f < nfields; f++) F.crm[f][idx] = F.gcm[f][t];
}

// modules::perturb_temperature  (pam_core/modules/perturb_temperature.h:10-63) with splitmix64 for yakl::Random (absent
// third-party generator; see the oracle).  One thread per (level < nz/4, member) walks its cells three times in the
// reference's serial order: mean before, perturb + mean after, rescale.
__device__ __forceinline__ double splitmix64_unit(unsigned long long seed) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
__global__ void __launch_bounds__(64) perturb_temperature_kernel(int nens, int nx, int ny, int num_levels, double *temp,
                                                                 const int *__restrict__ id, double magnitude) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)num_levels * nens) return;
  const int e = (int)(t % nens), k = (int)(t / nens);
  const double r_nx_ny = 1.0 / (nx * ny);
  const long long base = (long long)k * ny * nx * nens + e;
  const int ncol = ny * nx;
  double hmean1 = 0, hmean2 = 0;
  for (int c = 0; c < ncol; c++) hmean1 += temp[base + (long long)c * nens] * r_nx_ny;
  const long long seed0 = (long long)id[e] * num_levels * ny * nx + (long long)k * ny * nx;   // + j*nx + i = + c
  const double scaling = (num_levels - (double)k) / num_levels;
  for (int c = 0; c < ncol; c++) {
    double rnd = splitmix64_unit((unsigned long long)(seed0 + c)) * 2. - 1.;
    rnd = fmin(rnd, 1.0);
    rnd = fmax(rnd, -1.0);
    const double v = temp[base + (long long)c * nens] + rnd * magnitude * scaling;
    temp[base + (long long)c * nens] = v;
    hmean2 += v * r_nx_ny;
  }
  for (int c = 0; c < ncol; c++) {
    const long long o = base + (long long)c * nens;
    temp[o] = temp[o] * hmean1 / hmean2;
  }
}

}  // namespace

extern "C" int pam_amd_set_last_error_(int code, const char *msg);   // defined in awfl_kernels.hip

extern "C" int pam_amd_sponge_layer(int nens, int nx, int ny, int nz, int num_fields, double *const *fields,
                                    const double *zint, const double *zmid, double crm_dt, int num_layers, double time_scale,
                                    double *workspace, void *stream) {
  if (nens < 1 || nx < 1 || ny < 1 || nz < 1 || !fields || !zint || !zmid)
    return pam_amd_set_last_error_(PAM_AMD_EINVAL, "sponge_layer: bad dimensions or null pointer");
  if (num_fields < 5 || num_fields > MAX_FIELDS)
    return pam_amd_set_last_error_(PAM_AMD_EINVAL, "sponge_layer: num_fields must be 5 + number of tracers (<= 55)");
  if (num_layers < 1 || num_layers > nz)
    return pam_amd_set_last_error_(PAM_AMD_EINVAL, "sponge_layer: sponge_num_layers must be in [1, nz]");
  if (!(time_scale > 0)) return pam_amd_set_last_error_(PAM_AMD_EINVAL, "sponge_layer: sponge_time_scale must be positive");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return pam_amd_set_last_error_(PAM_AMD_ENOGPU, "sponge_layer: no HIP device available (this library has no CPU path)");
  FieldPtrs F;
  for (int i = 0; i < MAX_FIELDS; i++) F.p[i] = nullptr;
  for (int i = 0; i < num_fields; i++) {
    if (!fields[i]) return pam_amd_set_last_error_(PAM_AMD_EINVAL, "sponge_layer: null field pointer");
    F.p[i] = fields[i];
  }
  hipStream_t s = (hipStream_t)stream;
  (void)workspace;      // (the horizontal means live in the workgroups since ABI 5; the argument is kept for callers of ABI <= 4)
  const int ME = nens < 64 ? nens : 64;
  const dim3 grid((unsigned)((nens + ME - 1) / ME), (unsigned)num_layers, (unsigned)num_fields), block((unsigned)ME, (unsigned)MOD_NS);
  hipLaunchKernelGGL(sponge_kernel, grid, block, 0, s, F, nens, nx * ny, nz, num_layers, zint, zmid, crm_dt / time_scale);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return pam_amd_set_last_error_(PAM_AMD_ENOGPU, hipGetErrorString(err));
  return PAM_AMD_OK;
}

namespace {
int kessler_check(int nens, int nx, int ny, int nz, const void *a, const void *b, const void *c, const void *d, const void *e,
                  const void *f, const void *g, double dt, double R_d, double R_v, double cp_d, double p0) {
  if (nens < 1 || nx < 1 || ny < 1 || nz < 2) return pam_amd_set_last_error_(PAM_AMD_EINVAL, "kessler: bad dimensions (nz >= 2)");
  if (!a || !b || !c || !d || !e || !f || !g) return pam_amd_set_last_error_(PAM_AMD_EINVAL, "kessler: null pointer");
  if (!(dt > 0) || !(R_d > 0) || !(R_v > 0) || !(cp_d > 0) || !(p0 > 0))
    return pam_amd_set_last_error_(PAM_AMD_EINVAL, "kessler: dt and the constants must be positive");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return pam_amd_set_last_error_(PAM_AMD_ENOGPU, "kessler: no HIP device available (this library has no CPU path)");
  return PAM_AMD_OK;
}

// the tables of pow_pos_fast on the device that holds the caller's arrays (`ref`: any of them; built once per device and process, 3.5 KB;
// the FIRST call on a device allocates and copies synchronously -- pam_amd_modules_finalize() frees them).  The launches go to the
// caller's stream, which must belong to that device, as for any kernel launch.
std::mutex g_tab_mutex;
std::vector<PowTab *> g_tabs;
const PowTab *kessler_pow_tab(const void *ref) {
  int dev = -1, cur = -1;
  hipPointerAttribute_t attr;
  if (ref && hipPointerGetAttributes(&attr, ref) == hipSuccess) dev = attr.device;
  else (void)hipGetLastError();
  if (hipGetDevice(&cur) != hipSuccess) return nullptr;
  if (dev < 0) dev = cur;
  std::lock_guard<std::mutex> lk(g_tab_mutex);
  if ((int)g_tabs.size() <= dev) g_tabs.resize(dev + 1, nullptr);
  if (!g_tabs[dev]) {
    PowTab host;
    pama::build_pow_tab(host);
    PowTab *d = nullptr;
    if (dev != cur && hipSetDevice(dev) != hipSuccess) return nullptr;
    const bool ok = hipMalloc((void **)&d, sizeof(PowTab)) == hipSuccess && hipMemcpy(d, &host, sizeof(PowTab), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok && d) (void)hipFree(d);
    if (dev != cur) (void)hipSetDevice(cur);
    if (!ok) return nullptr;
    g_tabs[dev] = d;
  }
  return g_tabs[dev];
}

// grid of kessler_limit_kernel: 256 columns per workgroup; the levels are dealt to as few workgroups as still give ~4096 of them
dim3 kessler_limit_grid(long long ncol, int nz) {
  const long long nxb = (ncol + 255) / 256;
  long long nyb = (4096 + nxb - 1) / nxb;
  if (nyb > nz - 1) nyb = nz - 1;
  if (nyb < 1) nyb = 1;
  return dim3((unsigned)nxb, (unsigned)nyb);
}

int kessler_read_dt_max(const double *slot, hipStream_t s, double *out) {
  double v = 0;
  if (hipMemcpyAsync(&v, slot, sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
    return pam_amd_set_last_error_(PAM_AMD_ENOGPU, hipGetErrorString(hipGetLastError()));
  if (!(v > 0)) return pam_amd_set_last_error_(PAM_AMD_ESTATE, "kessler: sedimentation time-step limit is not positive (NaN or negative rain/density in the coupler state)");
  *out = v;
  return PAM_AMD_OK;
}
}  // namespace

extern "C" int pam_amd_modules_finalize(void) {
  std::lock_guard<std::mutex> lk(g_tab_mutex);
  int cur = -1;
  (void)hipGetDevice(&cur);
  for (size_t d = 0; d < g_tabs.size(); d++)
    if (g_tabs[d]) {
      if ((int)d != cur) (void)hipSetDevice((int)d);
      (void)hipFree(g_tabs[d]);
      g_tabs[d] = nullptr;
    }
  if (cur >= 0) (void)hipSetDevice(cur);
  return PAM_AMD_OK;
}

extern "C" int pam_amd_kessler_max_stable_dt(int nens, int nx, int ny, int nz, const double *rho_r, const double *rho_dry,
                                             const double *zmid, double dt, double *workspace, void *stream, double *dt_max) {
  if (!dt_max) return pam_amd_set_last_error_(PAM_AMD_EINVAL, "kessler: null dt_max");
  if (int rc = kessler_check(nens, nx, ny, nz, rho_r, rho_r, rho_r, rho_dry, rho_dry, zmid, workspace, dt, 1, 1, 1, 1)) return rc;
  const PowTab *tab = kessler_pow_tab(workspace);
  if (!tab) return pam_amd_set_last_error_(PAM_AMD_ENOMEM, "kessler: cannot allocate the pow tables");
  hipStream_t s = (hipStream_t)stream;
  const long long ncol = (long long)ny * nx * nens;
  unsigned long long *slot = (unsigned long long *)(workspace + (long long)nz * ncol);
  if (hipMemsetAsync(slot, 0x7f, 8, s) != hipSuccess) return pam_amd_set_last_error_(PAM_AMD_ENOGPU, hipGetErrorString(hipGetLastError()));
  hipLaunchKernelGGL(kessler_limit_kernel, kessler_limit_grid(ncol, nz), dim3(256), 0, s, nz, ncol, nens, rho_r, rho_dry, zmid, dt, slot, tab);
  return kessler_read_dt_max((const double *)slot, s, dt_max);
}

extern "C" int pam_amd_kessler_time_step(int nens, int nx, int ny, int nz, double *rho_v, double *rho_c, double *rho_r,
                                         const double *rho_dry, double *temp, double *precl, const double *zmid, double dt,
                                         double R_d, double R_v, double cp_d, double p0, double *workspace, void *stream,
                                         int rainsplit_hint, int *rainsplit) {
  if (!precl) return pam_amd_set_last_error_(PAM_AMD_EINVAL, "kessler: null precl");
  if (int rc = kessler_check(nens, nx, ny, nz, rho_v, rho_c, rho_r, rho_dry, temp, zmid, workspace, dt, R_d, R_v, cp_d, p0)) return rc;
  const PowTab *tab = kessler_pow_tab(workspace);
  if (!tab) return pam_amd_set_last_error_(PAM_AMD_ENOMEM, "kessler: cannot allocate the pow tables");
  hipStream_t s = (hipStream_t)stream;
  const long long ncol = (long long)ny * nx * nens;
  unsigned long long *slot = (unsigned long long *)(workspace + (long long)nz * ncol);
  int n = rainsplit_hint;
  if (n <= 0) {
    // The sub-cycle count comes from a global minimum (the reference's yakl::intrinsics::minval, :389-390): one 8-byte
    // read-back.  The kernel that takes it writes nothing, so that a failure here (a NaN state, an absurd sub-cycle count, a HIP
    // error) leaves the coupler arrays untouched.
    if (hipMemsetAsync(slot, 0x7f, 8, s) != hipSuccess) return pam_amd_set_last_error_(PAM_AMD_ENOGPU, hipGetErrorString(hipGetLastError()));
    hipLaunchKernelGGL(kessler_limit_kernel, kessler_limit_grid(ncol, nz), dim3(256), 0, s, nz, ncol, nens, rho_r, rho_dry, zmid, dt, slot, tab);
    double dt_max;
    if (int rc = kessler_read_dt_max((const double *)slot, s, &dt_max)) return rc;
    const double want = ceil(dt / dt_max);
    if (!(want < 1.e6)) return pam_amd_set_last_error_(PAM_AMD_ESTATE, "kessler: more than 1e6 sedimentation sub-cycles requested");
    n = (int)want;
    if (n < 1) n = 1;
  }
  const dim3 cgrid((unsigned)((ncol + 63) / 64)), cblock(64);
  const bool narrow = (long long)nz * ncol < (1ll << 29);
#define PAMA_KESSLER_COLUMN(SINGLE, IDX)                                                                                         \
  hipLaunchKernelGGL((kessler_column_kernel<SINGLE, IDX>), cgrid, cblock, 0, s, nz, ncol, nens, rho_v, rho_c, rho_r, rho_dry, temp, \
                     precl, zmid, workspace, dt, n, R_d, R_v, cp_d, p0, tab)
  if (n == 1 && narrow) PAMA_KESSLER_COLUMN(true, unsigned);
  else if (n == 1) PAMA_KESSLER_COLUMN(true, long long);
  else if (narrow) PAMA_KESSLER_COLUMN(false, unsigned);
  else PAMA_KESSLER_COLUMN(false, long long);
#undef PAMA_KESSLER_COLUMN
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return pam_amd_set_last_error_(PAM_AMD_ENOGPU, hipGetErrorString(err));
  if (rainsplit) *rainsplit = n;
  return PAM_AMD_OK;
}

namespace {
int gcm_check(const char *who, int nens, int nx, int ny, int nz, const void *const *a, int na, const void *const *b, int nb,
              const void *const *c, int nc) {
  if (nens < 1 || nx < 1 || ny < 1 || nz < 1 || !a || !b || !c)
    return pam_amd_set_last_error_(PAM_AMD_EINVAL, (std::string(who) + ": bad dimensions or null pointer table").c_str());
  for (int i = 0; i < na; i++) if (!a[i]) return pam_amd_set_last_error_(PAM_AMD_EINVAL, (std::string(who) + ": null CRM field pointer").c_str());
  for (int i = 0; i < nb; i++) if (!b[i]) return pam_amd_set_last_error_(PAM_AMD_EINVAL, (std::string(who) + ": null GCM column pointer").c_str());
  for (int i = 0; i < nc; i++) if (!c[i]) return pam_amd_set_last_error_(PAM_AMD_EINVAL, (std::string(who) + ": null tendency pointer").c_str());
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return pam_amd_set_last_error_(PAM_AMD_ENOGPU, (std::string(who) + ": no HIP device available (this library has no CPU path)").c_str());
  return PAM_AMD_OK;
}
}  // namespace

extern "C" int pam_amd_gcm_forcing_compute(int nens, int nx, int ny, int nz, const double *const *crm, const double *const *gcm,
                                           double *const *tend, double gcm_physics_dt, void *stream) {
  if (int rc = gcm_check("compute_gcm_forcing_tendencies", nens, nx, ny, nz, (const void *const *)crm, 10,
                         (const void *const *)gcm, 10, (const void *const *)tend, 14)) return rc;
  if (!(gcm_physics_dt > 0)) return pam_amd_set_last_error_(PAM_AMD_EINVAL, "compute_gcm_forcing_tendencies: gcm_physics_dt must be positive");
  Gcm10 C, G; Gcm14 T;
  for (int i = 0; i < 10; i++) { C.p[i] = const_cast<double *>(crm[i]); G.p[i] = const_cast<double *>(gcm[i]); }
  for (int i = 0; i < 14; i++) T.p[i] = tend[i];
  hipStream_t s = (hipStream_t)stream;
  const int ME = nens < 64 ? nens : 64;
  const dim3 grid((unsigned)((nens + ME - 1) / ME), (unsigned)nz), block((unsigned)ME, (unsigned)GCM_NS);
  const bool small = (long long)nz * ny * nx * nens < (1ll << 29);
  if (small) hipLaunchKernelGGL(gcm_forcing_compute_kernel<unsigned>, grid, block, 0, s, nens, nx * ny, nz, C, G, T, 1.0 / gcm_physics_dt);
  else hipLaunchKernelGGL(gcm_forcing_compute_kernel<long long>, grid, block, 0, s, nens, nx * ny, nz, C, G, T, 1.0 / gcm_physics_dt);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return pam_amd_set_last_error_(PAM_AMD_ENOGPU, hipGetErrorString(err));
  return PAM_AMD_OK;
}

extern "C" int pam_amd_gcm_forcing_apply(int nens, int nx, int ny, int nz, double *const *crm, const double *const *gcm,
                                         double *const *tend, const double *dz, double crm_dt, double gcm_physics_dt,
                                         double *workspace, void *stream, int *mask_out) {
  if (int rc = gcm_check("apply_gcm_forcing_tendencies", nens, nx, ny, nz, (const void *const *)crm, 10, (const void *const *)gcm,
                         10, (const void *const *)tend, 14)) return rc;
  if (!dz || !workspace) return pam_amd_set_last_error_(PAM_AMD_EINVAL, "apply_gcm_forcing_tendencies: null dz or workspace");
  if (!(gcm_physics_dt > 0)) return pam_amd_set_last_error_(PAM_AMD_EINVAL, "apply_gcm_forcing_tendencies: gcm_physics_dt must be positive");
  Gcm10 C, G; Gcm14 T;
  for (int i = 0; i < 10; i++) { C.p[i] = crm[i]; G.p[i] = const_cast<double *>(gcm[i]); }
  for (int i = 0; i < 14; i++) T.p[i] = tend[i];
  hipStream_t s = (hipStream_t)stream;
  const long long n2 = (long long)nz * nens, per_level = (long long)ny * nx * nens, ncell = per_level * nz;
  double *glob = workspace + 6 * n2;
  int *flags = (int *)(glob + 2 * (long long)nens);
  if (hipMemsetAsync(flags, 0, 8 * sizeof(int), s) != hipSuccess) return pam_amd_set_last_error_(PAM_AMD_ENOGPU, hipGetErrorString(hipGetLastError()));
  const int ME = nens < 64 ? nens : 64;
  const dim3 grid((unsigned)((nens + ME - 1) / ME), (unsigned)nz), block((unsigned)ME, (unsigned)GCM_NS_APPLY);
  if (ncell < (1ll << 29))
    hipLaunchKernelGGL(gcm_forcing_apply_kernel<unsigned>, grid, block, 0, s, nens, nx * ny, nz, C, G, T, dz, crm_dt, 1.0 / gcm_physics_dt,
                       workspace, flags);
  else
    hipLaunchKernelGGL(gcm_forcing_apply_kernel<long long>, grid, block, 0, s, nens, nx * ny, nz, C, G, T, dz, crm_dt, 1.0 / gcm_physics_dt,
                       workspace, flags);
  // "Only do the hole filling if there's negative mass" (:432-436) and ScalarLiveOut neg_too_large (:241,:252): one read-back
  int h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyAsync(h, flags, sizeof(h), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
    return pam_amd_set_last_error_(PAM_AMD_ENOGPU, hipGetErrorString(hipGetLastError()));
  int mask = 0;
  for (int sp = 0; sp < 3; sp++) {
    if (!h[sp]) continue;
    mask |= 1 << sp;
    double *rho_x = crm[GF_RV + sp];
    const double *neg = workspace + (long long)sp * n2, *pos = workspace + (long long)(3 + sp) * n2;
    hipLaunchKernelGGL(gcm_fill_level_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, s, nens, per_level, ncell, rho_x, dz,
                       neg, pos);
    if (h[3 + sp]) {
      mask |= 16 << sp;
      hipLaunchKernelGGL(gcm_fill_glob_sum_kernel, dim3((unsigned)((nens + 63) / 64)), dim3(64), 0, s, nens, nx, ny, nz, rho_x, dz, neg,
                         pos, glob);
      hipLaunchKernelGGL(gcm_fill_glob_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, s, nens, per_level, ncell, rho_x,
                         dz, glob);
    }
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return pam_amd_set_last_error_(PAM_AMD_ENOGPU, hipGetErrorString(err));
  if (mask_out) *mask_out = mask;
  return PAM_AMD_OK;
}

extern "C" int pam_amd_broadcast_initial_gcm_column(int nens, int nx, int ny, int nz, int num_fields, const double *const *gcm,
                                                    double *const *crm, void *stream) {
  if (nens < 1 || nx < 1 || ny < 1 || nz < 1 || !gcm || !crm)
    return pam_amd_set_last_error_(PAM_AMD_EINVAL, "broadcast_initial_gcm_column: bad dimensions or null pointer table");
  if (num_fields != 1 && num_fields != 6)
    return pam_amd_set_last_error_(PAM_AMD_EINVAL, "broadcast_initial_gcm_column: num_fields must be 6 (all) or 1 (dry density only)");
  Ptr6 F;
  for (int f = 0; f < 6; f++) { F.crm[f] = nullptr; F.gcm[f] = nullptr; }
  for (int f = 0; f < num_fields; f++) {
    if (!gcm[f] || !crm[f]) return pam_amd_set_last_error_(PAM_AMD_EINVAL, "broadcast_initial_gcm_column: null field pointer");
    F.crm[f] = crm[f]; F.gcm[f] = gcm[f];
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return pam_amd_set_last_error_(PAM_AMD_ENOGPU, "broadcast_initial_gcm_column: no HIP device available (this library has no CPU path)");
  const long long per_level = (long long)ny * nx * nens, ncell = per_level * nz;
  hipLaunchKernelGGL(broadcast_gcm_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, (hipStream_t)stream, nens, per_level,
                     ncell, num_fields, F);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return pam_amd_set_last_error_(PAM_AMD_ENOGPU, hipGetErrorString(err));
  return PAM_AMD_OK;
}

extern "C" int pam_amd_perturb_temperature(int nens, int nx, int ny, int nz, double *temp, const int *id, double magnitude,
                                           void *stream) {
  if (nens < 1 || nx < 1 || ny < 1 || nz < 1 || !temp || !id)
    return pam_amd_set_last_error_(PAM_AMD_EINVAL, "perturb_temperature: bad dimensions or null pointer");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return pam_amd_set_last_error_(PAM_AMD_ENOGPU, "perturb_temperature: no HIP device available (this library has no CPU path)");
  const int num_levels = nz / 4;
  if (num_levels == 0) return PAM_AMD_OK;
  const long long n = (long long)num_levels * nens;
  hipLaunchKernelGGL(perturb_temperature_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, nens, nx, ny,
                     num_levels, temp, id, magnitude);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return pam_amd_set_last_error_(PAM_AMD_ENOGPU, hipGetErrorString(err));
  return PAM_AMD_OK;
}


// ------------------------------------------------------------------------------------------------
// supercell_init (standalone/mmf_simplified/supercell_init.h:7-135): the standalone driver's supercell COLUMN -- dry density,
// winds, temperature and vapour density of each level, from the analytic sounding (supercell_sounding.h) with the total
// pressure integrated hydrostatically through 5 Gauss-Lobatto points per cell.  One workgroup (the column is nz ~ 60 levels):
//   phase 1  every (level, GLL interval) integrates -(1+qv) g / ((R_d + qv R_v) T) over its 5 sub-points   (:46-66,:76-79)
//   phase 2  one lane chains the exponentials from the ground up (the reference does the same in a 1-iteration kernel, :70-87)
//   phase 3  every level averages its 5 GLL points                                                           (:92-133)
namespace {
__global__ void __launch_bounds__(256) supercell_init_kernel(int nz, const double *__restrict__ zint, double R_d, double R_v,
                                                             double grav, double *__restrict__ rho_d_col,
                                                             double *__restrict__ uvel_col, double *__restrict__ vvel_col,
                                                             double *__restrict__ wvel_col, double *__restrict__ temp_col,
                                                             double *__restrict__ rho_v_col) {
  constexpr int ord = 5;
  const double gll_pts[ord] = {-0.50000000000000000000000000000000000000, -0.32732683535398857189914622812342917778,
                               0.00000000000000000000000000000000000000, 0.32732683535398857189914622812342917778,
                               0.50000000000000000000000000000000000000};
  const double gll_wts[ord] = {0.050000000000000000000000000000000000000, 0.27222222222222222222222222222222222222,
                               0.35555555555555555555555555555555555556, 0.27222222222222222222222222222222222222,
                               0.050000000000000000000000000000000000000};
  extern __shared__ double sc_lds[];
  double *tot = sc_lds;                       // (nz, ord-1): integral of the log-pressure gradient over each GLL interval
  double *hyp = sc_lds + (size_t)nz * (ord - 1);   // (nz, ord): total pressure at the GLL points
  const pama::Sounding snd = pama::Sounding::make(zint[nz], R_d, grav);
  for (int t = threadIdx.x; t < nz * (ord - 1); t += blockDim.x) {
    const int k = t / (ord - 1), kk = t - k * (ord - 1);
    const double dz = zint[k + 1] - zint[k];
    const double cellmid = zint[k] + 0.5 * dz;
    const double ord_b = cellmid + gll_pts[kk] * dz, ord_t = cellmid + gll_pts[kk + 1] * dz;
    const double ord_m = 0.5 * (ord_b + ord_t);
    const double ord_dz = dz * (gll_pts[kk + 1] - gll_pts[kk]);
    double acc = 0;
    for (int kkk = 0; kkk < ord; kkk++) {
      double temp;
      const double qv = snd.vapour_mixing_ratio(ord_m + ord_dz * gll_pts[kkk], temp);
      acc += (-(1 + qv) * grav / (R_d + qv * R_v) / temp) * gll_wts[kkk];
    }
    tot[t] = acc * (dz * (gll_pts[kk + 1] - gll_pts[kk]));
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    hyp[0] = snd.p_0;
    for (int k = 0; k < nz; k++)
      for (int kk = 0; kk < ord - 1; kk++) {
        hyp[k * ord + kk + 1] = hyp[k * ord + kk] * exp(tot[k * (ord - 1) + kk]);
        if (kk == ord - 2 && k < nz - 1) hyp[(k + 1) * ord] = hyp[k * ord + ord - 1];
      }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < nz; k += blockDim.x) {
    double rd = 0, u = 0, v = 0, w = 0, T = 0, rv = 0;
    const double dz = zint[k + 1] - zint[k];
    const double zmid = 0.5 * (zint[k] + zint[k + 1]);
    for (int kk = 0; kk < ord; kk++) {
      const double zloc = zmid + gll_pts[kk] * dz;
      double temp;
      const double qv = snd.vapour_mixing_ratio(zloc, temp);
      const double rho_d = hyp[k * ord + kk] / (R_d + qv * R_v) / temp;
      const double zs = 5000, us = 30, uc = 15;
      const double uvel = (zloc < zs) ? us * (zloc / zs) - uc : us - uc;
      rd += rho_d * gll_wts[kk];
      u += uvel * gll_wts[kk];
      v += 0.0 * gll_wts[kk];
      w += 0.0 * gll_wts[kk];
      T += temp * gll_wts[kk];
      rv += (qv * rho_d) * gll_wts[kk];
    }
    rho_d_col[k] = rd; uvel_col[k] = u; vvel_col[k] = v; wvel_col[k] = w; temp_col[k] = T; rho_v_col[k] = rv;
  }
}
}  // namespace

extern "C" int pam_amd_supercell_init(int nz, const double *vert_interface, double R_d, double R_v, double grav,
                                      double *rho_d_col, double *uvel_col, double *vvel_col, double *wvel_col, double *temp_col,
                                      double *rho_v_col, void *stream) {
  if (nz < 1 || !vert_interface || !rho_d_col || !uvel_col || !vvel_col || !wvel_col || !temp_col || !rho_v_col)
    return pam_amd_set_last_error_(PAM_AMD_EINVAL, "supercell_init: bad nz or null pointer");
  const size_t lds = (size_t)nz * 9 * sizeof(double);
  if (lds > 160 * 1024) return pam_amd_set_last_error_(PAM_AMD_EINVAL, "supercell_init: more than 2275 levels");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return pam_amd_set_last_error_(PAM_AMD_ENOGPU, "supercell_init: no HIP device available (this library has no CPU path)");
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute((const void *)supercell_init_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return pam_amd_set_last_error_(PAM_AMD_ENOGPU, "supercell_init: cannot raise the LDS limit");
  hipLaunchKernelGGL(supercell_init_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, nz, vert_interface, R_d, R_v, grav,
                     rho_d_col, uvel_col, vvel_col, wvel_col, temp_col, rho_v_col);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return pam_amd_set_last_error_(PAM_AMD_ENOGPU, hipGetErrorString(err));
  return PAM_AMD_OK;
}
