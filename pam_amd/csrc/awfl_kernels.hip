// awfl_kernels.hip -- gfx950 kernels, host orchestration and C ABI of the MI355X-native AWFL dycore step.
//
// Built by __graft_entry__.build():  hipcc --offload-arch=gfx950 -O3 -fPIC -shared ... -o libpam_amd_awfl.so
// Declared in include/pam_amd_awfl.h.  There is no CPU fallback: without a HIP device every compute entry point
// returns PAM_AMD_ENOGPU.
//
// Launch structure of one SSPRK3 sub-step (reference: 3 x Dycore::compute_tendencies + 3 combines, Dycore.h:147-222;
// ~18 launches and ~13 allocations there):
//     fused stage (default), per stage:
//                 awfl_flux_kernel<.,DIFF>  y and z sweeps in ONE launch (Dycore.h:387-519)
//                 awfl_xupd_kernel          x sweep + update of the state and of water vapour (its own FCT multiplier + the update an
//                                           unlimited neighbourhood gets; Dycore.h:334-386,525-584,162-221, next stage's :310-321 divide,
//                                           :662-710 ghosts); for large ensembles also phase 1 of the further tracers' x sweeps
//                 awfl_xtr_kernel<.,1>      (small ensembles) phase 1 of the further tracers: x fluxes -> their FCT multipliers
//                 awfl_xtr_kernel<.,2>      phase 2: the same x fluxes again + the complete limited update (NT > 1 only)
//                 awfl_ptail_kernel         next stage's pressure + density/pressure ghosts (Dycore.h:310-321,:682-709)
//                 awfl_trfix_kernel         water vapour redone where its limiter acted (driven by line flags; leaves at once elsewhere)
//     three-kernel stage (cross-check, every face flux and multiplier stored):
//                 awfl_flux_kernel (x, y, z) -> awfl_fct_kernel (Dycore.h:525-550) -> awfl_update_kernel
// All scratch is allocated once in init.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pam_amd_awfl.h"
#include "awfl_device.h"
#include "awfl_vertical.h"

using namespace pama;

// ------------------------------------------------------------------------------------------------ kernels
// grid: [0,nbx) x-sweep blocks, [nbx,nbx+nby) y-sweep, rest z-sweep.  Within a sweep every WAVEFRONT is one work
// unit: unit u -> (item block of 64 items = u / nspan, span index = u % nspan), i.e. consecutive wavefronts take
// consecutive spans of the same items.
// Ensemble sub-range [e0, e0+ne) processed by one launch (the arrays keep their full-nens strides).  Local thread
// indices are flattened over (cell or line, local member) with the member fastest, then mapped to global indices.
struct EnsRange { int e0, ne; };
__device__ __forceinline__ long long to_global(long long t, int nens, EnsRange R) {
  const long long r = t / R.ne;
  return r * nens + R.e0 + (t - r * R.ne);
}

// Pointwise kernels: grid (ceil(nx*ne/256), ny, nz); k and j come from the block indices, (i, local member) from one
// 32-bit division.  Small ensembles (P.flat_cells: nx*nens does not fill workgroups of 256): the grid is flat over every cell
// and a thread finds (k, j, i, e) with three 32-bit divisions.
__device__ __forceinline__ bool flat_cell(const Params &P, int nlev, CellId &c) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned sz = (unsigned)P.sz, sy = (unsigned)P.sy, ne = (unsigned)P.nens;
  const unsigned k = t / sz, r = t - k * sz;
  if (k >= (unsigned)nlev) return false;
  const unsigned j = r / sy, r2 = r - j * sy, i = r2 / ne;
  c.k = (int)k; c.j = (int)j; c.i = (int)i; c.e = (int)(r2 - i * ne);
  c.idx = (long long)t;
  return true;
}
__device__ __forceinline__ bool grid_cell(const Params &P, EnsRange R, CellId &c) {
  if (P.flat_cells) return flat_cell(P, P.nz, c);
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (unsigned)P.nx * (unsigned)R.ne) return false;
  const unsigned i = t / (unsigned)R.ne;
  c.k = blockIdx.z; c.j = blockIdx.y; c.i = (int)i; c.e = R.e0 + (int)(t - i * (unsigned)R.ne);
  c.idx = (((long long)c.k * P.ny + c.j) * P.nx + c.i) * P.nens + c.e;
  return true;
}
static inline dim3 cell_grid(const Params &P, EnsRange r, int nlev = 0) {
  if (nlev <= 0) nlev = P.nz;
  if (P.flat_cells) return dim3((unsigned)(((long long)nlev * P.sz + 255) / 256), 1, 1);
  return dim3((unsigned)(((long long)P.nx * r.ne + 255) / 256), (unsigned)P.ny, (unsigned)nlev);
}

struct FluxGrid {
  int nbx, nby, nbz;        // workgroups per sweep
  int spx, spy, spz;        // faces per thread (span) per sweep
  int nsx, nsy, nsz;        // spans per line
  int nux, nuy, nuz;        // work units (wavefronts) per sweep: lines x member blocks x spans
  int nbx_l, nby_l;         // x / y workgroups per vertical level when the two sweeps are interleaved level by level (else 0)
  int part;                 // -1: whole sweeps per wavefront; 0: pass 1 only; 1: one pair of advected fields per wavefront
  int npx, npy, npz;        // part == 1: pairs per sweep (the unit index carries the pair)
};

// DIFF (the fused stage's y/z sweeps): momentum and theta leave as per-cell flux differences (flux_line_body); there is no x
// sweep in that mode (the fused x-sweep kernel does it).
// FLAT (small ensembles, DIFF only): the lanes of a wavefront are 64 consecutive items of the sweep's flat index space
// ((level, x, member) for y, (y, x, member) for z: flat_lane) instead of 64 members of one line; G.nsy / G.nsz spans per item group.
// FOLD (DIFF, member lanes, 3-D): this launch is the z sweep of a folded stage -- the y sweep ran in the launch before -- and stores
// the y+z part of each state variable's divergence instead of its z difference (flux_line_body_zt<., FOLD>).
// VZ_PER_ENS here means the FLAT lanes of small ensembles with per-member vertical grids (each lane reads its member's table from
// global memory: 62 registers of coefficients per trip, so two wavefronts per SIMD instead of four -- no scratch); member lanes with
// per-member grids run awfl_fluxz_pe_kernel below.
template <bool VZ_PER_ENS, bool DIFF, bool FLAT, bool FOLD = false>
__global__ void __launch_bounds__(FLUX_THREADS, VZ_PER_ENS ? 2 : 4) awfl_flux_kernel(Params P, FluxGrid G, EnsRange R,
                                                                     const double *__restrict__ prim,
                                                                     double *__restrict__ fx, double *__restrict__ fy,
                                                                     double *__restrict__ fz) {
  static_assert(DIFF || !FLAT, "flat lanes exist for the fused stage's y/z sweeps only");
  static_assert(!FOLD || (DIFF && !FLAT), "the fold exists for the member-lane z sweep of the fused stage");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nblk = (R.ne + 63) >> 6;          // blocks of 64 members per line
  // Dispatch order: the z-sweep workgroups first, then the x/y sweeps.
  int b = (int)blockIdx.x - G.nbz;
  if (b < 0) b += G.nbx + G.nby + G.nbz;      // blockIdx < nbz  ->  logical index in [nbx+nby, nbx+nby+nbz)
  // x- and y-sweeps read the same horizontal slab of the state: when their workgroups tile the levels evenly they are
  // interleaved level by level, so that the second sweep of a level finds the slab in the Infinity Cache.
  if (G.nbx_l > 0 && b < G.nbx + G.nby) {
    const int per = G.nbx_l + G.nby_l;
    const int lev = b / per, r = b % per;
    b = (r < G.nbx_l) ? lev * G.nbx_l + r : G.nbx + lev * G.nby_l + (r - G.nbx_l);
  }
  // wave unit u of a sweep -> (group = u / nspan, span = u % nspan), group -> (line = group / nblk, member block)
  if (b < G.nbx) {
    const int u = b * FLUX_WAVES + wave;
    if (u < G.nux) {
      const int up = (G.part == 1) ? uni_int(u / G.npx) : u, psel = (G.part == 1) ? 1 + (u - up * G.npx) : G.part;
      const int grp = uni_int(up / G.nsx), line = uni_int(grp / nblk), el = (grp - line * nblk) * 64 + lane;
      if (!DIFF && el < R.ne)
        flux_line_body<0, VZ_PER_ENS, false>(P, prim, fx, line, R.e0 + el, (up - grp * G.nsx) * G.spx, G.spx, psel);
    }
  } else if (b < G.nbx + G.nby) {
    const int u = (b - G.nbx) * FLUX_WAVES + wave;
    if (u < G.nuy) {
      const int up = (G.part == 1) ? uni_int(u / G.npy) : u, psel = (G.part == 1) ? 1 + (u - up * G.npy) : G.part;
      const int grp = uni_int(up / G.nsy);
      if (FLAT) {
        const unsigned q = (unsigned)grp * 64u + (unsigned)lane;
        if (q < (unsigned)flat_items(P, 1))
          flux_line_body<1, VZ_PER_ENS, DIFF>(P, prim, fy, flat_lane<1>(P, q), (up - grp * G.nsy) * G.spy, G.spy, psel);
      } else {
        const int line = uni_int(grp / nblk), el = (grp - line * nblk) * 64 + lane;
        if (el < R.ne)
          flux_line_body<1, VZ_PER_ENS, DIFF>(P, prim, fy, line, R.e0 + el, (up - grp * G.nsy) * G.spy, G.spy, psel);
      }
    }
  } else {
    const int u = (b - G.nbx - G.nby) * FLUX_WAVES + wave;
    if (u < G.nuz) {
      const int up = (G.part == 1) ? uni_int(u / G.npz) : u, psel = (G.part == 1) ? 1 + (u - up * G.npz) : G.part;
      const int grp = uni_int(up / G.nsz);
      if (FLAT) {
        const unsigned q = (unsigned)grp * 64u + (unsigned)lane;
        if (q < (unsigned)flat_items(P, 2))
          flux_line_body<2, VZ_PER_ENS, DIFF>(P, prim, fz, flat_lane<2>(P, q), (up - grp * G.nsz) * G.spz, G.spz, psel);
      } else {
        const int line = uni_int(grp / nblk), el = (grp - line * nblk) * 64 + lane;
        if (el < R.ne)
          flux_line_body<2, VZ_PER_ENS, DIFF, FOLD>(P, prim, fz, line, R.e0 + el, (up - grp * G.nsz) * G.spz, G.spz, psel, fy);
      }
    }
  }
}

// The z sweep of ensembles whose members have DIFFERENT vertical grids (the coupler's general contract: set_grid(..., realConst2d),
// pam_coupler.h:163-181; the reference builds and applies the vertical WENO matrices per (k, iens) unconditionally, Dycore.h:897-940,
// :454-481), member lanes.  A level's table is 31 doubles per member: read by every lane for every polynomial it is 248 B against the
// 8 B of field data the polynomial consumes.  Here the ZW wavefronts of a workgroup sweep ZW neighbouring columns of the SAME block of
// 64 members, level by level in step, and the table of the level -- 31 x 64 doubles -- is staged in LDS once per workgroup (double
// buffered: the next level's rows are requested before the trip's polynomials and stored behind them, one barrier per level); each
// lane then reads its member's coefficients from LDS.  Workgroups are numbered member block by member block, so at any time the
// chip works on one or two blocks and their tables (62 levels x 15.9 KB ~ 1 MB) stay in every XCD's L2.
// Same weno5_table on the same values as ZTabLane: same bits.  Lanes / wavefronts beyond the ensemble range / the last column are
// clamped to the last valid member / column and redo its work (identical values to identical addresses) -- every lane of the
// workgroup reaches every barrier and helps staging.
#ifndef PAMA_ZPE_WAVES
#define PAMA_ZPE_WAVES 4
#endif
constexpr int ZPE_WAVES = PAMA_ZPE_WAVES;
struct ZTabLds {
  double *buf;              // LDS: two tables of VZ_STRIDE x 64 doubles
  const double *vz;         // (nz + 2, VZ_STRIDE, nens)
  long long nens;
  int eb, emax;             // first member of the block, last valid member
  int tid, lane, last;
  double r[(VZ_STRIDE * 64 + 64 * ZPE_WAVES - 1) / (64 * ZPE_WAVES)];
  static constexpr int NR = (VZ_STRIDE * 64 + 64 * ZPE_WAVES - 1) / (64 * ZPE_WAVES);
  __device__ __forceinline__ double rdz(const Params &P, int k, int e) const { return P.rdz[(long long)k * P.nens + e]; }
  __device__ __forceinline__ void load(int level) {
    const double *src = vz + (long long)level * VZ_STRIDE * nens;
#pragma unroll
    for (int i = 0; i < NR; i++) {
      const int idx = tid + i * 64 * ZPE_WAVES;            // row m = idx / 64 of the table, member eb + idx % 64
      const int m = idx >> 6, l = idx & 63, em = (eb + l < emax) ? eb + l : emax;
      r[i] = (idx < VZ_STRIDE * 64) ? src[(long long)m * nens + em] : 0.0;
    }
  }
  __device__ __forceinline__ void store(int level) {
    double *dst = buf + (level & 1) * (VZ_STRIDE * 64);
#pragma unroll
    for (int i = 0; i < NR; i++) {
      const int idx = tid + i * 64 * ZPE_WAVES;
      if (idx < VZ_STRIDE * 64) dst[idx] = r[i];
    }
  }
  __device__ __forceinline__ void pass_begin(int first, int last_) {
    last = last_;
    load(first);
    store(first);
    __syncthreads();
  }
  __device__ __forceinline__ void trip_begin(int level) { if (level < last) load(level + 1); }
  __device__ __forceinline__ void trip_end(int level) {
    if (level < last) store(level + 1);
    __syncthreads();
  }
  __device__ __forceinline__ void weno(const double u[5], int level, const WenoConsts &wc, double &L, double &R) const {
    weno5_table(u, buf + (level & 1) * (VZ_STRIDE * 64) + lane, 64, wc, L, R);
  }
};
// grid: blocks of member-block-major units: b -> (member block, [pair of advected fields], span, group of ZW columns)
// (two wavefronts per SIMD: the 62 registers of a level's coefficients stay live across the two or three polynomials of a trip -- 229-243
// registers, no scratch; with the budget of three wavefronts per SIMD (168) the compiler spills 336 B per lane)
template <bool DIFF, bool FOLD>
__global__ void __launch_bounds__(64 * ZPE_WAVES, 2) awfl_fluxz_pe_kernel(Params P, EnsRange R, int spz, int nsz, int part, int npz,
                                                                              const double *__restrict__ prim, const double *__restrict__ fy,
                                                                              double *__restrict__ fz) {
  __shared__ double ztab[2 * VZ_STRIDE * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int ncol = P.ny * P.nx, ncg = (ncol + ZPE_WAVES - 1) / ZPE_WAVES;
  const int per_blk = ncg * nsz * (part == 1 ? npz : 1);
  const int b = (int)blockIdx.x;
  const int blk = uni_int(b / per_blk), r1 = b - blk * per_blk;
  const int r2 = uni_int(r1 / ncg), cg = r1 - r2 * ncg;
  const int pp = uni_int(r2 / nsz), sp = r2 - pp * nsz;
  const int psel = (part == 1) ? 1 + pp : part;
  int line = cg * ZPE_WAVES + wave;
  if (line > ncol - 1) line = ncol - 1;
  const int emax = R.e0 + R.ne - 1;
  int e = R.e0 + blk * 64 + lane;
  if (e > emax) e = emax;
  ZTabLds zt;
  zt.buf = ztab; zt.vz = P.vz; zt.nens = P.nens; zt.eb = R.e0 + blk * 64; zt.emax = emax;
  zt.tid = (int)threadIdx.x; zt.lane = lane; zt.last = 0;
  flux_line_body_zt<2, DIFF, ZTabLds, FOLD>(P, prim, fz, member_lane<2>(P, line, e), sp * spz, spz, psel, zt, fy);
}

__global__ void __launch_bounds__(256) awfl_fct_kernel(Params P, EnsRange R, const double *__restrict__ fx,
                                                       const double *__restrict__ fy, const double *__restrict__ fz,
                                                       const double *__restrict__ seed, double *__restrict__ mult,
                                                       FctRows rows, double dt, int t0) {
  fct_rows_resolve(rows);
  CellId c;
  if (grid_cell(P, R, c)) fct_mult_body(P, fx, fy, fz, seed, mult, rows, dt, c, t0);
}
template <int STAGE>
__global__ void __launch_bounds__(256) awfl_update_kernel(Params P, EnsRange R, const double *prim_in,
                                                          const double *prim0, double *prim_out,
                                                          const double *__restrict__ fx, const double *__restrict__ fy,
                                                          const double *__restrict__ fz, const double *__restrict__ mult,
                                                          FctRows rows, double *__restrict__ seed, double dt_dyn) {
  fct_rows_resolve(rows);
  CellId c;
  if (grid_cell(P, R, c)) update_body<STAGE>(P, prim_in, prim0, prim_out, fx, fy, fz, mult, rows, seed, dt_dyn, c);
}
// Fused x-sweep + state update (flux_x_update_body): wave unit u -> (x line, member block, span of cells); the 64 lanes are
// 64 consecutive members of ONE line, so every address is a wave-uniform base + member (scalar addressing).
template <int STAGE, bool FOLD>
__global__ void __launch_bounds__(FLUX_THREADS, 2) awfl_xupd_kernel(Params P, EnsRange R, const double *__restrict__ prim_in,
                                                                     const double *__restrict__ prim0,
                                                                     double *__restrict__ prim_out, double *__restrict__ fx,
                                                                     const double *__restrict__ fy,
                                                                     const double *__restrict__ fz, double *__restrict__ seed,
                                                                     double *__restrict__ mult, FctRows rows, double dt_dyn,
                                                                     double dt_stage, int tracers_inline, int span, int nspan) {
  fct_rows_resolve(rows);
  const int u = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * FLUX_WAVES + (threadIdx.x >> 6)));
  const int nblk = (R.ne + 63) >> 6;
  const int grp = uni_int(u / nspan), sp = u - grp * nspan;
  const int line = uni_int(grp / nblk), blk = grp - line * nblk;
  const int el = blk * 64 + (int)(threadIdx.x & 63);
  if (line < P.nz * P.ny && el < R.ne)
    flux_x_update_body<STAGE, FOLD>(P, prim_in, prim0, prim_out, fx, fy, fz, seed, mult, rows, line, R.e0 + el, sp * span, span,
                                    dt_dyn, dt_stage, tracers_inline != 0);
}
// x sweeps of tracers 1.. (x_tracer_sweep): wave unit u -> (x line, member block, span, pair of tracers).  PHASE 1 (the cells' FCT
// multipliers; only for small ensembles -- otherwise it runs inline in awfl_xupd_kernel) and PHASE 2 (the cells' complete update)
// both follow awfl_xupd_kernel (they need the face mass flux; phase 2 also the new density and every line's multipliers).
template <int STAGE, int PHASE, bool AHEAD = false>
__device__ __forceinline__ void xtr_pairs_block(const Params &P, EnsRange R, const double *__restrict__ prim_in,
                                                const double *__restrict__ prim0, double *__restrict__ prim_out,
                                                const double *__restrict__ fx, const double *__restrict__ fy,
                                                const double *__restrict__ fz, double *__restrict__ seed,
                                                double *__restrict__ mult, const FctRows &rows, double dt_dyn,
                                                double dt_stage, int npairs, int span, int nspan, int block) {
  const int u = __builtin_amdgcn_readfirstlane((int)(block * FLUX_WAVES + (threadIdx.x >> 6)));
  const int nblk = (R.ne + 63) >> 6;
  const int g2 = uni_int(u / npairs), pair = u - g2 * npairs;
  const int grp = uni_int(g2 / nspan), sp = g2 - grp * nspan;
  const int line = uni_int(grp / nblk), el = (grp - line * nblk) * 64 + (int)(threadIdx.x & 63);
  if (line < P.nz * P.ny && el < R.ne) {
    // advected-field indices of the pair's tracers (water vapour rides with the state pass)
    const int fa[2] = {4 + further_tracer(P, 2 * pair), 4 + further_tracer(P, 2 * pair + 1)};
    if (2 * pair + 1 < P.nt - 1)
      x_tracer_sweep<2, STAGE, PHASE, AHEAD>(P, prim_in, prim0, prim_out, fx, fy, fz, seed, mult, rows, line, R.e0 + el, sp * span, span, fa, dt_dyn, dt_stage, false, 0.0);
    else
      x_tracer_sweep<1, STAGE, PHASE, AHEAD>(P, prim_in, prim0, prim_out, fx, fy, fz, seed, mult, rows, line, R.e0 + el, sp * span, span, fa, dt_dyn, dt_stage, false, 0.0);
  }
}
template <int STAGE, int PHASE, bool AHEAD = false>
__global__ void __launch_bounds__(FLUX_THREADS) awfl_xtr_kernel(Params P, EnsRange R, const double *__restrict__ prim_in,
                                                               const double *__restrict__ prim0, double *__restrict__ prim_out,
                                                               const double *__restrict__ fx, const double *__restrict__ fy,
                                                               const double *__restrict__ fz, double *__restrict__ seed,
                                                               double *__restrict__ mult, FctRows rows, double dt_dyn,
                                                               double dt_stage, int npairs, int span, int nspan) {
  fct_rows_resolve(rows);
  xtr_pairs_block<STAGE, PHASE, AHEAD>(P, R, prim_in, prim0, prim_out, fx, fy, fz, seed, mult, rows, dt_dyn, dt_stage, npairs, span, nspan,
                                       (int)blockIdx.x);
}
// The same sweeps with G = 4 or G = 1 further tracers per wavefront (pam_amd_awfl_set_tracer_grouping; experiment (a) of VERDICT r3 / r4
// and its opposite).  Four: half the wavefronts, the per-wavefront loads a pair shares (face mass flux, the three densities) shared by
// four tracers -- and twice the windows in registers.  One: twice the wavefronts, each with half the registers (more wavefronts per
// SIMD), the shared loads repeated per tracer.  Kernels of their own so that the pair form keeps its register count.  Same arithmetic
// per tracer: same bits.
template <int STAGE, int PHASE, int G>
__device__ __forceinline__ void xtr_groups_block(const Params &P, EnsRange R, const double *__restrict__ prim_in,
                                                 const double *__restrict__ prim0, double *__restrict__ prim_out,
                                                 const double *__restrict__ fx, const double *__restrict__ fy,
                                                 const double *__restrict__ fz, double *__restrict__ seed,
                                                 double *__restrict__ mult, const FctRows &rows, double dt_dyn,
                                                 double dt_stage, int ngroups, int span, int nspan, int block) {
  static_assert(G == 1 || G == 4, "groups of one or four tracers");
  const int u = __builtin_amdgcn_readfirstlane((int)(block * FLUX_WAVES + (threadIdx.x >> 6)));
  const int nblk = (R.ne + 63) >> 6;
  const int g2 = uni_int(u / ngroups), grpn = u - g2 * ngroups;
  const int grp = uni_int(g2 / nspan), sp = g2 - grp * nspan;
  const int line = uni_int(grp / nblk), el = (grp - line * nblk) * 64 + (int)(threadIdx.x & 63);
  if (line < P.nz * P.ny && el < R.ne) {
    const int first = G * grpn, left = P.nt - 1 - first;
    const int fa[4] = {4 + further_tracer(P, first), 4 + further_tracer(P, first + 1), 4 + further_tracer(P, first + 2), 4 + further_tracer(P, first + 3)};
    if (G == 4 && left >= 4)
      x_tracer_sweep<4, STAGE, PHASE>(P, prim_in, prim0, prim_out, fx, fy, fz, seed, mult, rows, line, R.e0 + el, sp * span, span, fa, dt_dyn, dt_stage, false, 0.0);
    else if (G == 4 && left == 3)
      x_tracer_sweep<3, STAGE, PHASE>(P, prim_in, prim0, prim_out, fx, fy, fz, seed, mult, rows, line, R.e0 + el, sp * span, span, fa, dt_dyn, dt_stage, false, 0.0);
    else if (G == 4 && left == 2)
      x_tracer_sweep<2, STAGE, PHASE>(P, prim_in, prim0, prim_out, fx, fy, fz, seed, mult, rows, line, R.e0 + el, sp * span, span, fa, dt_dyn, dt_stage, false, 0.0);
    else
      x_tracer_sweep<1, STAGE, PHASE>(P, prim_in, prim0, prim_out, fx, fy, fz, seed, mult, rows, line, R.e0 + el, sp * span, span, fa, dt_dyn, dt_stage, false, 0.0);
  }
}
template <int STAGE, int PHASE, int G>
__global__ void __launch_bounds__(FLUX_THREADS) awfl_xtrn_kernel(Params P, EnsRange R, const double *__restrict__ prim_in,
                                                                const double *__restrict__ prim0, double *__restrict__ prim_out,
                                                                const double *__restrict__ fx, const double *__restrict__ fy,
                                                                const double *__restrict__ fz, double *__restrict__ seed,
                                                                double *__restrict__ mult, FctRows rows, double dt_dyn,
                                                                double dt_stage, int ngroups, int span, int nspan) {
  fct_rows_resolve(rows);
  xtr_groups_block<STAGE, PHASE, G>(P, R, prim_in, prim0, prim_out, fx, fy, fz, seed, mult, rows, dt_dyn, dt_stage, ngroups, span, nspan,
                                    (int)blockIdx.x);
}
// Pointwise tail of the fused stage, (1): next stage's pressure + density/pressure ghosts: a pow per cell and nothing else, so few
// registers and full occupancy; TAIL_LEVELS levels per thread (a sixth of the wavefronts, the (i, member) split once).
constexpr int TAIL_LEVELS = 6;
// the tables of pow_pos_fast (3.5 KB) staged in LDS: five per-lane lookups per pow, which are what the pressure pass waits for when
// they go to global memory (0.39 -> 0.26 ms on C2)
__device__ __forceinline__ void stage_pow_tab(Params &P, PowTab *sh_tab) {
  const double *src = reinterpret_cast<const double *>(P.pw);
  double *dst = reinterpret_cast<double *>(sh_tab);
  for (int i = threadIdx.x; i < (int)(sizeof(PowTab) / sizeof(double)); i += blockDim.x) dst[i] = src[i];
  __syncthreads();
  P.pw = sh_tab;
}
// member-lane grid (ceil(nx*ne/256), ny, level groups): the block (bx, by, bz) of that grid
__device__ __forceinline__ void ptail_block(const Params &P, EnsRange R, double *__restrict__ prim_out, unsigned bx, int by, int bz) {
  const unsigned t = bx * blockDim.x + threadIdx.x;
  if (t >= (unsigned)P.nx * (unsigned)R.ne) return;
  CellId c;
  const unsigned i = t / (unsigned)R.ne;
  c.j = by; c.i = (int)i; c.e = R.e0 + (int)(t - i * (unsigned)R.ne);
#pragma unroll
  for (int kk = 0; kk < TAIL_LEVELS; kk++) {
    c.k = bz * TAIL_LEVELS + kk;
    if (c.k >= P.nz) return;
    c.idx = (((long long)c.k * P.ny + c.j) * P.nx + c.i) * P.nens + c.e;
    pressure_tail_body(P, prim_out, c);
  }
}
__global__ void __launch_bounds__(256) awfl_ptail_kernel(Params P, EnsRange R, double *__restrict__ prim_out) {
  __shared__ PowTab sh_tab;
  stage_pow_tab(P, &sh_tab);
  if (!P.flat_cells) { ptail_block(P, R, prim_out, blockIdx.x, (int)blockIdx.y, (int)blockIdx.z); return; }
  CellId c;
  const int ngrp = (P.nz + TAIL_LEVELS - 1) / TAIL_LEVELS;       // groups of levels
  if (!flat_cell(P, ngrp, c)) return;
  const int kg = c.k;
#pragma unroll
  for (int kk = 0; kk < TAIL_LEVELS; kk++) {
    c.k = kg * TAIL_LEVELS + kk;
    if (c.k >= P.nz) return;
    c.idx = (((long long)c.k * P.ny + c.j) * P.nx + c.i) * P.nens + c.e;
    pressure_tail_body(P, prim_out, c);
  }
}
// (2) the fix-up of water vapour (the tracer finished in the state pass) where the limiter acted (tracer_fixup_line_body): wave unit u -> (x line, member block).  Every
// wavefront leaves after ONE scalar load unless some row (of any tracer) was flagged in this stage, and after its five line flags
// unless a row of its own or of a neighbouring line was.
template <int STAGE>
__device__ __forceinline__ void trfix_block(const Params &P, EnsRange R, const double *__restrict__ prim_in,
                                            const double *__restrict__ prim0, double *prim_out,
                                            const double *__restrict__ fx, const double *__restrict__ fy,
                                            const double *__restrict__ fz, const double *__restrict__ mult,
                                            const FctRows &rows, double *__restrict__ seed, double dt_dyn, int block) {
  const int u = __builtin_amdgcn_readfirstlane((int)(block * 4 + (threadIdx.x >> 6)));
  const int nblk = (R.ne + 63) >> 6;
  const int line = uni_int(u / nblk), el = (u - line * nblk) * 64 + (int)(threadIdx.x & 63);
  if (line >= P.nz * P.ny || el >= R.ne) return;
  if (rows.any[(R.e0 + el) >> 6] != rows.seq) return;     // no row of this member block was flagged in this stage
  const int k = uni_int(line / P.ny), j = line - k * P.ny;
  tracer_fixup_line_body<STAGE>(P, prim_in, prim0, prim_out, fx, fy, fz, mult, rows, seed, dt_dyn, P.idWV, k, j, R.e0 + el);
}
template <int STAGE>
__global__ void __launch_bounds__(256) awfl_trfix_kernel(Params P, EnsRange R, const double *__restrict__ prim_in,
                                                         const double *__restrict__ prim0, double *prim_out,
                                                         const double *__restrict__ fx, const double *__restrict__ fy,
                                                         const double *__restrict__ fz, const double *__restrict__ mult,
                                                         FctRows rows, double *__restrict__ seed, double dt_dyn) {
  fct_rows_resolve(rows);
  trfix_block<STAGE>(P, R, prim_in, prim0, prim_out, fx, fy, fz, mult, rows, seed, dt_dyn, (int)blockIdx.x);
}
// NT > 1, member-lane sweeps: the LAST three launches of a stage in one -- phase 2 of the further tracers' x sweeps (SINGLES: one tracer
// per wavefront, else pairs), the pressure pass and water vapour's fix-up.  They read what the x-sweep and phase 1 wrote and write
// disjoint things (the further tracers + their seeds; the pressure + density / pressure ghosts; water vapour where its limiter acted), so
// nothing orders them among themselves: the long phase-2 workgroups are dispatched first, the short pointwise ones fill its tail.  Same
// bodies, same bits; two launch boundaries and the idle ends of two short launches less per stage (C4 shard: ptail 15 + fix-up 6 us of a
// 420 us stage).
//   grid: [0, nb_xtr) phase 2 | [nb_xtr, nb_xtr + nb_pt) pressure pass, nb_pt = ptx * ny * level groups | rest: fix-up
template <int STAGE, bool SINGLES>
__global__ void __launch_bounds__(FLUX_THREADS) awfl_xtr2_tail_kernel(Params P, EnsRange R, const double *__restrict__ prim_in,
                                                                     const double *__restrict__ prim0, double *prim_out,
                                                                     const double *__restrict__ fx, const double *__restrict__ fy,
                                                                     const double *__restrict__ fz, double *__restrict__ seed,
                                                                     double *__restrict__ mult, FctRows rows, FctRows rows_fix, double dt_dyn,
                                                                     double dt_stage, int ngroups, int span, int nspan, int nb_xtr, int nb_pt,
                                                                     int ptx) {
  static_assert(FLUX_THREADS == 256, "the three bodies share workgroups of 256 lanes");
  __shared__ PowTab sh_tab;
  const int b = (int)blockIdx.x;
  if (b < nb_xtr) {
    fct_rows_resolve(rows);
    if (SINGLES) xtr_groups_block<STAGE, 2, 1>(P, R, prim_in, prim0, prim_out, fx, fy, fz, seed, mult, rows, dt_dyn, dt_stage, ngroups, span, nspan, b);
    else xtr_pairs_block<STAGE, 2>(P, R, prim_in, prim0, prim_out, fx, fy, fz, seed, mult, rows, dt_dyn, dt_stage, ngroups, span, nspan, b);
  } else if (b < nb_xtr + nb_pt) {
    stage_pow_tab(P, &sh_tab);
    const int q = b - nb_xtr, per_level = ptx * P.ny;
    const int bz = q / per_level, r = q - bz * per_level, by = r / ptx;
    ptail_block(P, R, prim_out, (unsigned)(r - by * ptx), by, bz);
  } else {
    fct_rows_resolve(rows_fix);
    trfix_block<STAGE>(P, R, prim_in, prim0, prim_out, fx, fy, fz, mult, rows_fix, seed, dt_dyn, b - nb_xtr - nb_pt);
  }
}
// the same fix-up with one lane per cell (small ensembles; tracer_fixup_cell_body)
template <int STAGE>
__global__ void __launch_bounds__(256) awfl_trfix_flat_kernel(Params P, const double *__restrict__ prim_in,
                                                              const double *__restrict__ prim0, double *prim_out,
                                                              const double *__restrict__ fx, const double *__restrict__ fy,
                                                              const double *__restrict__ fz, const double *__restrict__ mult,
                                                              FctRows rows, double *__restrict__ seed, double dt_dyn) {
  fct_rows_resolve(rows);
  CellId c;
  if (!flat_cell(P, P.nz, c)) return;
  if (rows.any[c.e >> 6] != rows.seq) return;     // no row of this member block was flagged in this stage
  tracer_fixup_cell_body<STAGE>(P, prim_in, prim0, prim_out, fx, fy, fz, mult, rows, seed, dt_dyn, P.idWV, c);
}

// TILE form of the further tracers' x sweeps (one pair of tracers); have_ruf: inline in the state kernel (phase 1 only).
// SHUF: a whole line of the tile lies inside one wavefront (xtile_line_in_wavefront): the stencil values, the right-edge values and the
// face fluxes come from the neighbouring lanes by wavefront shuffles -- no LDS image, no workgroup barrier (XShuf, awfl_device.h)
template <int STAGE, int PHASE, int NF, bool SHUF>
__device__ __forceinline__ void xtr_tile_run(const Params &P, const XLane &X, const XShuf &S, int T, int TS, const double *__restrict__ prim_in,
                                             const double *__restrict__ prim0, double *__restrict__ prim_out,
                                             const double *__restrict__ fx, const double *__restrict__ fy,
                                             const double *__restrict__ fz, double *__restrict__ seed, double *__restrict__ mult,
                                             const FctRows &rows, double dt_dyn, double dt_stage, const int *fa, double *lds,
                                             bool have_ruf, double ruf_reg) {
  double L[NF], R[NF], cen[NF], F[NF], own[NF];
  int fields[NF];
#pragma unroll
  for (int n = 0; n < NF; n++) fields[n] = P_U + fa[n];
  if (SHUF) {
    xtile_load_own<NF>(P, prim_in, X, fields, own);
    if (X.poly) {      // (whole-line tiles: poly == face == upd; the lanes a lane shuffles with are lanes of its own line)
      xtile_tracer_polys_from<NF>(P, [&](int n, int s) { return xtile_shfl(own[n], S.ln[s]); }, own, L, R, cen);
#pragma unroll
      for (int f = 0; f < NF; f++) R[f] = xtile_shfl(R[f], S.l);
      xtile_tracer_face<NF>(P, fx, X, L, R, F, have_ruf, ruf_reg);
      double Fhi[NF];
#pragma unroll
      for (int f = 0; f < NF; f++) Fhi[f] = xtile_shfl(F[f], S.r);
      xtile_tracer_finish<NF, STAGE, PHASE>(P, prim_in, prim0, prim_out, fy, fz, seed, mult, rows, X, fa, F, Fhi, cen, dt_dyn, dt_stage);
    }
    return;
  }
  // lds: NF staged fields of TS elements (later: the face fluxes), then NF x T right-edge values
  double *st = lds, *ex = lds + NF * TS;
  xtile_stage<NF>(P, prim_in, X, fields, st, TS, own);
  __syncthreads();
  if (X.poly) {
    xtile_tracer_polys<NF>(P, X, st, TS, own, L, R, cen);
#pragma unroll
    for (int f = 0; f < NF; f++) ex[f * T + X.slot] = R[f];
  }
  __syncthreads();
  if (X.face) {
#pragma unroll
    for (int f = 0; f < NF; f++) R[f] = ex[f * T + X.slot_l];
    xtile_tracer_face<NF>(P, fx, X, L, R, F, have_ruf, ruf_reg);
#pragma unroll
    for (int f = 0; f < NF; f++) st[f * T + X.slot] = F[f];
  }
  __syncthreads();
  if (X.upd) {
    double Fhi[NF];
#pragma unroll
    for (int f = 0; f < NF; f++) Fhi[f] = st[f * T + X.slot_r];
    xtile_tracer_finish<NF, STAGE, PHASE>(P, prim_in, prim0, prim_out, fy, fz, seed, mult, rows, X, fa, F, Fhi, cen, dt_dyn, dt_stage);
  }
}
// Phase 1 of one tracer pair in a workgroup of its OWN inside the state kernel's launch (small grids with idle CUs: the state kernel's
// z slices 1.. ; round 5).  Inline, the pairs' phase 1 runs behind the state pass of the same wavefront -- one serial chain; here it runs
// BESIDE it, and instead of waiting for the state pass's face mass flux (another workgroup of the same launch) it forms that flux
// itself: the polynomials of rho*u and p of the lane's cell (the first two of xtile_state_polys: same products, same weno5_const) and
// acoustic_face on them -- the same values in the same functions, hence the same bits as the flux the state workgroup computes.
template <int STAGE, int NF, bool SHUF>
__device__ __forceinline__ void xtr_tile_run_own_flux(const Params &P, const XLane &X, const XShuf &S, int T, int TS,
                                                      const double *__restrict__ prim_in, const double *__restrict__ prim0,
                                                      double *__restrict__ prim_out, const double *__restrict__ fx,
                                                      const double *__restrict__ fy, const double *__restrict__ fz,
                                                      double *__restrict__ seed, double *__restrict__ mult, const FctRows &rows,
                                                      double dt_dyn, double dt_stage, const int *fa, double *lds) {
  constexpr int NQ = NF + 3;                         // staged: rho, p, u, the tracers
  constexpr int NE = NF + 2;                         // exchanged right-edge values: rho*u, p, the tracers
  const WenoConsts wc = weno_consts();
  int fields[NQ];
  fields[0] = P_RHO; fields[1] = P_PRES; fields[2] = P_U;
#pragma unroll
  for (int n = 0; n < NF; n++) fields[3 + n] = P_U + fa[n];
  double own[NQ], L[NE], R[NE], cen[NF], F[NF];
  double *st = lds, *ex = lds + NQ * TS;
  if (SHUF) {
    xtile_load_own<NQ>(P, prim_in, X, fields, own);
  } else {
    xtile_stage<NQ>(P, prim_in, X, fields, st, TS, own);
    __syncthreads();
  }
  auto nb = [&](int f, int s) -> double { return SHUF ? xtile_shfl(own[f], S.ln[s]) : st[f * TS + X.s5[s]]; };
  if (X.poly) {
    double r[5], u[5], w[5];
#pragma unroll
    for (int s = 0; s < 5; s++) { r[s] = (s == 2) ? own[0] : nb(0, s); u[s] = (s == 2) ? own[2] : nb(2, s); }
#pragma unroll
    for (int s = 0; s < 5; s++) w[s] = mul_rn(r[s], u[s]);
    weno5_const(w, wc, L[0], R[0]);
#pragma unroll
    for (int s = 0; s < 5; s++) w[s] = (s == 2) ? own[1] : nb(1, s);
    weno5_const(w, wc, L[1], R[1]);
#pragma unroll
    for (int n = 0; n < NF; n++) {
#pragma unroll
      for (int s = 0; s < 5; s++) w[s] = (s == 2) ? own[3 + n] : nb(3 + n, s);
      cen[n] = w[2];
      weno5_const(w, wc, L[2 + n], R[2 + n]);
    }
    if (!SHUF) {
#pragma unroll
      for (int f = 0; f < NE; f++) ex[f * T + X.slot] = R[f];
    }
  }
  if (!SHUF) __syncthreads();
  if (X.face) {
    double Rl[NE];
#pragma unroll
    for (int f = 0; f < NE; f++) Rl[f] = SHUF ? xtile_shfl(R[f], S.l) : ex[f * T + X.slot_l];
    double ruf, ppf;
    acoustic_face(Rl[0], L[0], Rl[1], L[1], false, ruf, ppf);
    double Lt[NF], Rt[NF];
#pragma unroll
    for (int n = 0; n < NF; n++) { Lt[n] = L[2 + n]; Rt[n] = Rl[2 + n]; }
    xtile_tracer_face<NF>(P, fx, X, Lt, Rt, F, true, ruf);
    if (!SHUF) {
#pragma unroll
      for (int f = 0; f < NF; f++) st[f * T + X.slot] = F[f];
    }
  }
  if (!SHUF) __syncthreads();
  if (X.upd) {
    double Fhi[NF];
#pragma unroll
    for (int f = 0; f < NF; f++) Fhi[f] = SHUF ? xtile_shfl(F[f], S.r) : st[f * T + X.slot_r];
    xtile_tracer_finish<NF, STAGE, 1>(P, prim_in, prim0, prim_out, fy, fz, seed, mult, rows, X, fa, F, Fhi, cen, dt_dyn, dt_stage);
  }
}
// The state pass of a tile (or one PART of it: XP_U / XP_VW / XP_T, awfl_device.h) for the lane's cell; F: the fluxes through its left face
template <int STAGE, bool SHUF, int PART>
__device__ __forceinline__ void xupd_tile_state(const Params &P, const XLane &X, const XShuf &S, int T, int TS, const double *__restrict__ prim_in,
                                                const double *__restrict__ prim0, double *__restrict__ prim_out, double *__restrict__ fx,
                                                const double *__restrict__ fy, const double *__restrict__ fz, double *__restrict__ seed,
                                                double *__restrict__ mult, const FctRows &rows, double dt_dyn, double dt_stage,
                                                bool with_pressure, double *xt_lds, double (&F)[XT_NF]) {
  double L[XT_NS], R[XT_NS], cen[6], own[XT_NS];
  int fields[XT_NS];
  xtile_state_fields(P, fields);
  const unsigned fmask = xtile_part_fields(PART);
  if (SHUF) {
    xtile_load_own<XT_NS>(P, prim_in, X, fields, own, fmask);
    if (X.poly) {
      xtile_state_polys_from<PART>(P, [&](int f, int s) { return xtile_shfl(own[f], S.ln[s]); }, own, L, R, cen);
#pragma unroll
      for (int f = 0; f < XT_NS; f++) R[f] = xtile_shfl(R[f], S.l);       // now: the right-edge values of the cell to the left
      xtile_state_face<PART>(P, fx, X, L, R, X.upd, F);
      double Fhi[XT_NF];
#pragma unroll
      for (int f = 0; f < XT_NF; f++) Fhi[f] = xtile_shfl(F[f], S.r);
      xtile_state_finish<STAGE, PART>(P, prim_in, prim0, prim_out, fy, fz, seed, mult, rows, X, F, Fhi, cen, dt_dyn, dt_stage, with_pressure);
    }
  } else {
    // LDS: XT_NS staged fields of TS elements each (later reused for the face fluxes), then XT_NS x T right-edge values
    double *st = xt_lds, *ex = xt_lds + XT_NS * TS;
    xtile_stage<XT_NS>(P, prim_in, X, fields, st, TS, own, fmask);
    __syncthreads();
    if (X.poly) {
      xtile_state_polys_from<PART>(P, [&](int f, int s) { return st[f * TS + X.s5[s]]; }, own, L, R, cen);
#pragma unroll
      for (int f = 0; f < XT_NS; f++) ex[f * T + X.slot] = R[f];
    }
    __syncthreads();
    if (X.face) {
#pragma unroll
      for (int f = 0; f < XT_NS; f++) R[f] = ex[f * T + X.slot_l];       // now: the right-edge values of the cell to the left
      xtile_state_face<PART>(P, fx, X, L, R, X.upd, F);
#pragma unroll
      for (int f = 0; f < XT_NF; f++) st[f * T + X.slot] = F[f];          // (every stencil read of the staged tile is behind the barrier)
    }
    __syncthreads();
    if (X.upd) {
      double Fhi[XT_NF];
#pragma unroll
      for (int f = 0; f < XT_NF; f++) Fhi[f] = st[f * T + X.slot_r];
      xtile_state_finish<STAGE, PART>(P, prim_in, prim0, prim_out, fy, fz, seed, mult, rows, X, F, Fhi, cen, dt_dyn, dt_stage, with_pressure);
    }
  }
}
// TILE form of the fused x-sweep (xtile_* in awfl_device.h): a lane per cell, right-edge values and face fluxes exchanged through
// LDS (XT_NS doubles per lane, used twice) -- or, SHUF, by wavefront shuffles when a line lies inside one wavefront.
// grid (tiles per line x member blocks, groups of lines), block (W, rows, lines per group).
template <int STAGE, bool SHUF>
__global__ void __launch_bounds__(1024) awfl_xupd_tile_kernel(Params P, XTileGeom G, const double *__restrict__ prim_in,
                                                              const double *__restrict__ prim0, double *__restrict__ prim_out,
                                                              double *__restrict__ fx, const double *__restrict__ fy,
                                                              const double *__restrict__ fz, double *__restrict__ seed,
                                                              double *__restrict__ mult, FctRows rows, double dt_dyn,
                                                              double dt_stage, int with_pressure, int tracers_inline, int state_parts) {
  fct_rows_resolve(rows);
  extern __shared__ double xt_lds[];
  const int T = (int)(blockDim.x * blockDim.y * blockDim.z);
  const int tid = (int)((threadIdx.z * blockDim.y + threadIdx.y) * blockDim.x + threadIdx.x);
  const XLane X = xtile_lane(P, G, (int)blockIdx.x, (int)blockIdx.y, (int)threadIdx.x, (int)threadIdx.y, (int)threadIdx.z);
  const int TS = xtile_stage_elems(G);
  // (round 5, measured and removed -- commit 38c84c8: the tables of pow_pos_fast staged in LDS for the pressure pass inside this kernel,
  // as awfl_ptail_kernel does: 0.628 -> 0.621 G at nens = 1, 0.106 -> 0.108 G on the 250 x 1 x 50 shape -- the staging and its barrier
  // cost what the nine chained look-ups of a boundary-level lane save)
  XShuf S = {};
  if (SHUF) S = xtile_shuffle_lanes(P, G, tid & 63, (int)threadIdx.x, (int)threadIdx.y);
  const int nz_state = state_parts ? 3 : 1;
  if ((int)blockIdx.z >= nz_state) {
    // tracers_inline == 2: the z slices behind the state pass's are phase 1 of the pairs of further tracers, beside the state pass
    const int i = 2 * ((int)blockIdx.z - nz_state);
    const int fa[2] = {4 + further_tracer(P, i), 4 + further_tracer(P, i + 1)};
    if (i + 1 < P.nt - 1)
      xtr_tile_run_own_flux<STAGE, 2, SHUF>(P, X, S, T, TS, prim_in, prim0, prim_out, fx, fy, fz, seed, mult, rows, dt_dyn, dt_stage, fa, xt_lds);
    else
      xtr_tile_run_own_flux<STAGE, 1, SHUF>(P, X, S, T, TS, prim_in, prim0, prim_out, fx, fy, fz, seed, mult, rows, dt_dyn, dt_stage, fa, xt_lds);
    return;
  }
  double F[XT_NF];
  if (state_parts) {          // the state pass in three parts beside each other (z slices 0 .. 2; xtile_state_* PART)
    if (blockIdx.z == 0) xupd_tile_state<STAGE, SHUF, XP_U>(P, X, S, T, TS, prim_in, prim0, prim_out, fx, fy, fz, seed, mult, rows, dt_dyn, dt_stage, with_pressure != 0, xt_lds, F);
    else if (blockIdx.z == 1) xupd_tile_state<STAGE, SHUF, XP_VW>(P, X, S, T, TS, prim_in, prim0, prim_out, fx, fy, fz, seed, mult, rows, dt_dyn, dt_stage, with_pressure != 0, xt_lds, F);
    else xupd_tile_state<STAGE, SHUF, XP_T>(P, X, S, T, TS, prim_in, prim0, prim_out, fx, fy, fz, seed, mult, rows, dt_dyn, dt_stage, with_pressure != 0, xt_lds, F);
    return;
  }
  xupd_tile_state<STAGE, SHUF, XP_ALL>(P, X, S, T, TS, prim_in, prim0, prim_out, fx, fy, fz, seed, mult, rows, dt_dyn, dt_stage, with_pressure != 0, xt_lds, F);
  // phase 1 of the further tracers (their FCT multipliers) inline -- small ensembles, where a launch costs more than the work: the
  // mass flux through the lane's left face is still in its register
  if (tracers_inline == 1) {
    for (int i = 0; i < P.nt - 1; i += 2) {
      const int fa[2] = {4 + further_tracer(P, i), 4 + further_tracer(P, i + 1)};
      if (!SHUF) __syncthreads();
      if (i + 1 < P.nt - 1)
        xtr_tile_run<STAGE, 1, 2, SHUF>(P, X, S, T, TS, prim_in, prim0, prim_out, fx, fy, fz, seed, mult, rows, dt_dyn, dt_stage, fa, xt_lds, true, F[0]);
      else
        xtr_tile_run<STAGE, 1, 1, SHUF>(P, X, S, T, TS, prim_in, prim0, prim_out, fx, fy, fz, seed, mult, rows, dt_dyn, dt_stage, fa, xt_lds, true, F[0]);
    }
  }
}
template <int STAGE, int PHASE, bool SHUF>
__global__ void __launch_bounds__(1024) awfl_xtr_tile_kernel(Params P, XTileGeom G, const double *__restrict__ prim_in,
                                                             const double *__restrict__ prim0, double *__restrict__ prim_out,
                                                             const double *__restrict__ fx, const double *__restrict__ fy,
                                                             const double *__restrict__ fz, double *__restrict__ seed,
                                                             double *__restrict__ mult, FctRows rows, double dt_dyn,
                                                             double dt_stage, int fixup_slice) {
  fct_rows_resolve(rows);
  extern __shared__ double xt_lds[];
  const int pair = (int)blockIdx.z;
  const int T = (int)(blockDim.x * blockDim.y * blockDim.z);
  const XLane X = xtile_lane(P, G, (int)blockIdx.x, (int)blockIdx.y, (int)threadIdx.x, (int)threadIdx.y, (int)threadIdx.z);
  if (PHASE == 2 && fixup_slice && pair == (int)gridDim.z - 1) {
    // the last z slice of a phase-2 launch of a small ensemble: water vapour's fix-up (tracer_fixup_cell_body), a lane per cell of
    // the tile -- everything it reads was written by earlier launches; one launch less per stage than awfl_trfix_flat_kernel
    if (X.upd && rows.any[X.e >> 6] == rows.seq) {
      CellId c;
      c.k = X.k; c.j = X.j; c.i = X.i; c.e = X.e; c.idx = (long long)X.io;
      tracer_fixup_cell_body<STAGE>(P, prim_in, prim0, prim_out, fx, fy, fz, mult, rows, seed, dt_dyn, P.idWV, c);
    }
    return;
  }
  const int TS = xtile_stage_elems(G);
  XShuf S = {};
  if (SHUF) S = xtile_shuffle_lanes(P, G, (int)((threadIdx.z * blockDim.y + threadIdx.y) * blockDim.x + threadIdx.x) & 63, (int)threadIdx.x, (int)threadIdx.y);
  const int fa[2] = {4 + further_tracer(P, 2 * pair), 4 + further_tracer(P, 2 * pair + 1)};
  if (2 * pair + 1 < P.nt - 1)
    xtr_tile_run<STAGE, PHASE, 2, SHUF>(P, X, S, T, TS, prim_in, prim0, prim_out, fx, fy, fz, seed, mult, rows, dt_dyn, dt_stage, fa, xt_lds, false, 0.0);
  else
    xtr_tile_run<STAGE, PHASE, 1, SHUF>(P, X, S, T, TS, prim_in, prim0, prim_out, fx, fy, fz, seed, mult, rows, dt_dyn, dt_stage, fa, xt_lds, false, 0.0);
}

// TILE form of the fused stage's y and z sweeps (ftile_* in awfl_device.h): a lane per cell, one launch for both directions --
// workgroups [0, nby) are y tiles, the rest z tiles.  1-D workgroups of T lanes (the larger of the two tile sizes).  LDS per lane:
// two ping-pong sets of FT_NG right-edge values + 4 face fluxes (the state variables' differences).
constexpr int FT_MAXG = (4 + MAXT + FT_NG - 1) / FT_NG + 1;
struct FTileGroups { int ny_groups, nz_groups; int gy[FT_MAXG][FT_NG], gz[FT_MAXG][FT_NG]; };
template <int DIR, bool VZ_PER_ENS>
__device__ __forceinline__ void flux_tile_run(const Params &P, const FTileGeom &G, int bx, int by, int T, const int (*grp)[FT_NG],
                                              int ngroups, const double *__restrict__ prim, double *__restrict__ flux, double *lds) {
  const int rows = ftile_rows(G), per = rows * G.W;
  const int t = (int)threadIdx.x;
  const int tz = t / per, r = t - tz * per, ty = r / G.W, tx = r - ty * G.W;
  FLane X = ftile_lane<DIR>(P, G, bx, by, tx, ty, tz < G.lpb ? tz : 0);
  if (tz >= G.lpb) { X.poly = X.face = X.own = X.pay = false; X.slot = X.slot_l = X.slot_r = 0; }
  double *ldsR[2] = {lds, lds + FT_NG * T}, *ldsF = lds + 2 * FT_NG * T;
  const int ncomp_a = (DIR == 1) ? 1 : 2;
  double L[FT_NG], R[FT_NG], F[FT_NG], ruf = 0.0, fn = 0.0;
  // (round 5, measured and removed -- commit 38c84c8: the stencils of group g + 1 requested before the polynomials of group g are
  // built, in a 512-lane instance with 143 instead of 113 registers: flux tile kernel 18.0 -> 21.8 us at nens = 1, 0.63 -> 0.57 G)
  if (X.poly) {
    ftile_acoustic_polys<DIR, VZ_PER_ENS>(P, prim, X, L, R);
#pragma unroll
    for (int n = 0; n < FT_NG; n++) ldsR[0][n * T + X.slot] = R[n];
  }
  __syncthreads();
  if (X.face) {
#pragma unroll
    for (int n = 0; n < FT_NG; n++) R[n] = ldsR[0][n * T + X.slot_l];
    ftile_acoustic_face<DIR>(P, flux, X, L, R, ruf, fn);
    ldsF[X.slot] = fn;
  }
  for (int g = 0; g < ngroups; g++) {
    double *buf = ldsR[(g + 1) & 1];
    int fa[FT_NG], nf = 0;
#pragma unroll
    for (int n = 0; n < FT_NG; n++) { fa[n] = grp[g][n]; nf += (fa[n] >= 0) ? 1 : 0; }
    if (X.poly) {
      ftile_adv_polys<DIR, VZ_PER_ENS>(P, prim, X, fa, nf, L, R);
#pragma unroll
      for (int n = 0; n < FT_NG; n++) buf[n * T + X.slot] = R[n];
    }
    __syncthreads();
    if (X.face) {
#pragma unroll
      for (int n = 0; n < FT_NG; n++) R[n] = buf[n * T + X.slot_l];
      ftile_adv_face<DIR>(P, flux, X, fa, nf, L, R, ruf, F);
    }
    if (g == 0) {              // every state variable is in group 0 of the advected ones: close the cells' flux differences
      if (X.face) {
#pragma unroll
        for (int n = 0; n < FT_NG; n++) ldsF[(1 + n) * T + X.slot] = F[n];
      }
      __syncthreads();
      if (X.pay) {
        ftile_store_diff<DIR>(P, flux, X, ncomp_a, fn, ldsF[X.slot_r]);
#pragma unroll
        for (int n = 0; n < FT_NG; n++)
          if (fa[n] >= 0 && fa[n] < 4) ftile_store_diff<DIR>(P, flux, X, fa[n], F[n], ldsF[(1 + n) * T + X.slot_r]);
      }
    }
  }
}
// The same tile with its parts BESIDE each other instead of behind each other (small grids with idle CUs; round 5): `part` 0 = the
// acoustic triple (face mass flux, stored; normal-momentum difference), part g + 1 = group g of the advected quantities in a workgroup
// of its own, which rebuilds the face mass flux from the polynomials of rho*u_n and p -- the same products and the same weno /
// acoustic_face as part 0: the same bits -- instead of waiting for it.  One exchange per workgroup instead of one per group in a chain.
template <int DIR, bool VZ_PER_ENS>
__device__ __forceinline__ void flux_tile_part(const Params &P, const FTileGeom &G, int bx, int by, int T, const int (*grp)[FT_NG],
                                               int ngroups, int part, const double *__restrict__ prim, double *__restrict__ flux,
                                               double *lds) {
  if (part > ngroups) return;
  const int rows = ftile_rows(G), per = rows * G.W;
  const int t = (int)threadIdx.x;
  const int tz = t / per, r = t - tz * per, ty = r / G.W, tx = r - ty * G.W;
  FLane X = ftile_lane<DIR>(P, G, bx, by, tx, ty, tz < G.lpb ? tz : 0);
  if (tz >= G.lpb) { X.poly = X.face = X.own = X.pay = false; X.slot = X.slot_l = X.slot_r = 0; }
  double *ldsA = lds, *ldsR = lds + FT_NG * T, *ldsF = lds + 2 * FT_NG * T;
  const int ncomp_a = (DIR == 1) ? 1 : 2;
  double La[FT_NG], Ra[FT_NG], L[FT_NG], R[FT_NG], F[FT_NG], ruf = 0.0, fn = 0.0;
  if (part == 0) {
    if (X.poly) {
      ftile_acoustic_polys<DIR, VZ_PER_ENS>(P, prim, X, La, Ra);
#pragma unroll
      for (int n = 0; n < FT_NG; n++) ldsA[n * T + X.slot] = Ra[n];
    }
    __syncthreads();
    if (X.face) {
#pragma unroll
      for (int n = 0; n < FT_NG; n++) Ra[n] = ldsA[n * T + X.slot_l];
      ftile_acoustic_face<DIR>(P, flux, X, La, Ra, ruf, fn);
      ldsF[X.slot] = fn;
    }
    __syncthreads();
    if (X.pay) ftile_store_diff<DIR>(P, flux, X, ncomp_a, fn, ldsF[X.slot_r]);
    return;
  }
  const int g = part - 1;
  int fa[FT_NG], nf = 0;
  bool has_state = false;
#pragma unroll
  for (int n = 0; n < FT_NG; n++) { fa[n] = grp[g][n]; nf += (fa[n] >= 0) ? 1 : 0; has_state |= (fa[n] >= 0 && fa[n] < 4); }
  if (X.poly) {
    ftile_acoustic_polys<DIR, VZ_PER_ENS, false>(P, prim, X, La, Ra);
    ftile_adv_polys<DIR, VZ_PER_ENS>(P, prim, X, fa, nf, L, R);
#pragma unroll
    for (int n = 0; n < 2; n++) ldsA[n * T + X.slot] = Ra[n];
#pragma unroll
    for (int n = 0; n < FT_NG; n++) ldsR[n * T + X.slot] = R[n];
  }
  __syncthreads();
  if (X.face) {
    double ppf;
    acoustic_face(ldsA[X.slot_l], La[0], ldsA[T + X.slot_l], La[1], (DIR == 2) && X.wall, ruf, ppf);     // (as ftile_acoustic_face)
#pragma unroll
    for (int n = 0; n < FT_NG; n++) R[n] = ldsR[n * T + X.slot_l];
    ftile_adv_face<DIR>(P, flux, X, fa, nf, L, R, ruf, F);
    if (has_state) {
#pragma unroll
      for (int n = 0; n < FT_NG; n++) ldsF[n * T + X.slot] = F[n];
    }
  }
  if (has_state) {            // (wave-uniform: the group is the workgroup's)
    __syncthreads();
    if (X.pay) {
#pragma unroll
      for (int n = 0; n < FT_NG; n++)
        if (fa[n] >= 0 && fa[n] < 4) ftile_store_diff<DIR>(P, flux, X, fa[n], F[n], ldsF[n * T + X.slot_r]);
    }
  }
}
// parts: 0 = every part of a tile in one workgroup, one behind the other; 1 = grid.y indexes the part (flux_tile_part)
template <bool VZ_PER_ENS>
__global__ void __launch_bounds__(VZ_PER_ENS ? 512 : 1024) awfl_flux_tile_kernel(Params P, FTileGeom Gy, FTileGeom Gz, FTileGroups Q, int nby, int gyx,
                                                              const double *__restrict__ prim, double *__restrict__ fy,
                                                              double *__restrict__ fz, int parts) {
  extern __shared__ double ft_lds[];
  const int T = (int)blockDim.x, b = (int)blockIdx.x;
  if (parts) {
    if (b < nby) flux_tile_part<1, VZ_PER_ENS>(P, Gy, b % gyx, b / gyx, T, Q.gy, Q.ny_groups, (int)blockIdx.y, prim, fy, ft_lds);
    else flux_tile_part<2, VZ_PER_ENS>(P, Gz, b - nby, 0, T, Q.gz, Q.nz_groups, (int)blockIdx.y, prim, fz, ft_lds);
    return;
  }
  if (b < nby) flux_tile_run<1, VZ_PER_ENS>(P, Gy, b % gyx, b / gyx, T, Q.gy, Q.ny_groups, prim, fy, ft_lds);
  else flux_tile_run<2, VZ_PER_ENS>(P, Gz, b - nby, 0, T, Q.gz, Q.nz_groups, prim, fz, ft_lds);
}

// Test hook: the device WENO arithmetic on its own (v_rcp_f64 + Newton reciprocals, FMA contraction, difference form).
// level < 0: uniform-grid constants (weno5_const, the x/y sweeps); else the per-level table `level` of member 0
// (weno5_table, the z sweep; level = vertical matrix index 0..nz+1 as in Dycore.h:454-469).
__global__ void __launch_bounds__(256) awfl_weno_kat_kernel(Params P, int level, const double *__restrict__ stencils, int n,
                                                            double *__restrict__ left, double *__restrict__ right) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const WenoConsts wc = weno_consts();
  double u[5], L, R;
  for (int s = 0; s < 5; s++) u[s] = stencils[(long long)t * 5 + s];
  if (level < 0) weno5_const(u, wc, L, R);
  else if (P.vz_per_ens) weno5_table(u, P.vz + ((long long)level * VZ_STRIDE) * P.nens, P.nens, wc, L, R);
  else weno5_table(u, P.vz + (long long)level * VZ_STRIDE, 1, wc, L, R);
  left[t] = L;
  right[t] = R;
}

__global__ void __launch_bounds__(256) awfl_init_thermal_kernel(Params P, EnsRange R, double xlen, double ylen, double cp_d,
                                                                double p0, const double *__restrict__ zmid,
                                                                double *__restrict__ rho_d, double *__restrict__ u,
                                                                double *__restrict__ v, double *__restrict__ w,
                                                                double *__restrict__ temp, TracerPtrs trc) {
  CellId c;
  if (grid_cell(P, R, c)) init_thermal_body(P, xlen, ylen, cp_d, p0, zmid, rho_d, u, v, w, temp, trc, c);
}

__global__ void __launch_bounds__(256) awfl_init_supercell_kernel(Params P, EnsRange R, const double *__restrict__ zmid,
                                                                  const double *__restrict__ hy_dens,
                                                                  const double *__restrict__ hy_pres,
                                                                  const double *__restrict__ dens_vap_gll,
                                                                  double *__restrict__ rho_d, double *__restrict__ u,
                                                                  double *__restrict__ v, double *__restrict__ w,
                                                                  double *__restrict__ temp, TracerPtrs trc) {
  CellId c;
  if (grid_cell(P, R, c)) init_supercell_body(P, zmid, hy_dens, hy_pres, dens_vap_gll, rho_d, u, v, w, temp, trc, c);
}

struct GcmPtrs { const double *p[5]; int use; };

__global__ void __launch_bounds__(256) awfl_init_prim_kernel(Params P, EnsRange R, const double *__restrict__ rho_d,
                                                             const double *__restrict__ u, const double *__restrict__ v,
                                                             const double *__restrict__ w, const double *__restrict__ temp,
                                                             TracerPtrs trc, GcmPtrs gcm, double *__restrict__ prim,
                                                             double *__restrict__ seed, int subtract_hy) {
  CellId c;
  if (grid_cell(P, R, c))
    init_prim_body(P, rho_d, u, v, w, temp, trc, gcm.use ? gcm.p : nullptr, prim, seed, subtract_hy != 0, c);
}

__global__ void __launch_bounds__(256) awfl_finalize_kernel(Params P, EnsRange R, const double *__restrict__ prim,
                                                            const double *__restrict__ seed, double *__restrict__ rho_d,
                                                            double *__restrict__ u, double *__restrict__ v,
                                                            double *__restrict__ w, double *__restrict__ temp,
                                                            TracerPtrs trc) {
  CellId c;
  if (grid_cell(P, R, c)) finalize_body(P, prim, seed, rho_d, u, v, w, temp, trc, c);
}

__global__ void __launch_bounds__(256) awfl_c2d_arrays_kernel(Params P, EnsRange R, const double *__restrict__ rho_d,
                                                              const double *__restrict__ u, const double *__restrict__ v,
                                                              const double *__restrict__ w, const double *__restrict__ temp,
                                                              TracerPtrs trc, double *__restrict__ state, double *__restrict__ tracers) {
  CellId c;
  if (grid_cell(P, R, c)) coupler_to_halo_arrays_body(P, rho_d, u, v, w, temp, trc, state, tracers, c);
}
__global__ void __launch_bounds__(256) awfl_d2c_arrays_kernel(Params P, EnsRange R, const double *__restrict__ state,
                                                              const double *__restrict__ tracers, double *__restrict__ rho_d,
                                                              double *__restrict__ u, double *__restrict__ v, double *__restrict__ w,
                                                              double *__restrict__ temp, TracerPtrs trc) {
  CellId c;
  if (grid_cell(P, R, c)) halo_arrays_to_coupler_body(P, state, tracers, rho_d, u, v, w, temp, trc, c);
}

// CFL reduction (Dycore.h:86-101): grid-stride min, wavefront shuffle reduce, the workgroup's four values through LDS, ONE atomicMin
// per workgroup on the bit pattern (positive doubles order like unsigned integers) -- and only when the value undercuts what the slot
// already holds: every wavefront of a launch updating one L2 address one after the other was ~90 us of this kernel at every size
// (100 us of a 10 ms step at one GPU's shard of C4, where all 8192 wavefronts are resident at once and finish together).
// LEAN (a step of one member range, where the host's wait is on the critical path of every step): no copy in front of or behind the
// kernel -- it puts +inf into the word the NEXT reduction will use (the host alternates between two) and the workgroup that finishes
// last publishes the minimum in the host's pinned word.  Otherwise the word is set and read back by copies on the stream (measured: with
// two member ranges the copy-free form is 1 % SLOWER at one GPU's shard of C4, nine alternating runs each -- not understood, kept apart).
template <bool LEAN>
__global__ void __launch_bounds__(256) awfl_cfl_kernel(Params P, const double *__restrict__ rho_d,
                                                       const double *__restrict__ u, const double *__restrict__ v,
                                                       const double *__restrict__ w, const double *__restrict__ temp,
                                                       const double *__restrict__ rho_v, double cfl,
                                                       unsigned long long *result, unsigned long long *next_result,
                                                       unsigned *done_count, unsigned long long *host_result) {
  __shared__ double wave_min[4];
  if (LEAN && blockIdx.x == 0 && threadIdx.x == 0) *next_result = 0x7FF0000000000000ull;
  double m = INFINITY;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < P.ncell;
       idx += (long long)gridDim.x * blockDim.x)
  {
    const double dtc = cfl_body(P, rho_d, u, v, w, temp, rho_v, cfl, idx);
    m = (dtc != dtc || m != m) ? NAN : fmin(m, dtc);     // fmin() drops a NaN operand: keep it, one bad cell must fail the step
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_down(m, off, 64);
    m = (o != o || m != m) ? NAN : fmin(m, o);
  }
  if ((threadIdx.x & 63) == 0) wave_min[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int wv = 1; wv < 4; wv++) {
      const double o = wave_min[wv];
      m = (o != o || m != m) ? NAN : fmin(m, o);
    }
    if (!(m > 0.0)) m = 0.0;   // NaN or non-positive anywhere in the workgroup's cells: reported as 0 (the host refuses it)
    const unsigned long long b = (unsigned long long)__double_as_longlong(m);
    if (b < __hip_atomic_load(result, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(result, b);
    if (!LEAN) return;
    // the workgroup that finishes last publishes the minimum in the host's pinned word and clears the count
    __threadfence();
    if (atomicAdd(done_count, 1u) == gridDim.x - 1) {
      __threadfence();
      *host_result = __hip_atomic_load(result, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *done_count = 0;
      __threadfence_system();
    }
  }
}

// The reference's runtime self-check (Dycore.h:36-58 compute_mass, used under PAM_DEBUG at :136-138, :224-251): per variable -- every
// tracer density, rho, rho*theta -- and per member the mean of q * dz over the member's cells, taken from the resident state (prim: rho,
// theta; seed: the conserved tracer densities).  grid (blocks of 64 members of the range, variables); lanes = members; the four
// wavefronts of a workgroup take every fourth column of each level and their partial sums are added in a fixed order: deterministic.
__global__ void __launch_bounds__(256) awfl_mass_kernel(Params P, EnsRange R, const double *__restrict__ prim,
                                                        const double *__restrict__ seed, double *__restrict__ mass) {
  __shared__ double part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int el = (int)blockIdx.x * 64 + lane;
  const bool ok = el < R.ne;
  const int e = R.e0 + (ok ? el : R.ne - 1);
  const int ivar = (int)blockIdx.y;
  const long long ncol = (long long)P.ny * P.nx;
  double s = 0.0;
  for (int k = 0; k < P.nz; k++) {
    const double dzk = P.dz[(long long)k * P.nens + e];
    for (long long c = wave; c < ncol; c += 4) {
      const long long o = (long long)(k + HS) * P.sz + c * P.nens + e;
      double q;
      if (ivar < P.nt) q = seed[(long long)ivar * P.ncell + ((long long)k * ncol + c) * P.nens + e];
      else if (ivar == P.nt) q = prim[P_RHO * P.prim_fs + o];
      else q = prim[P_RHO * P.prim_fs + o] * prim[P_THETA * P.prim_fs + o];
      s = fma(q, dzk, s);
    }
  }
  part[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && ok)
    mass[(long long)ivar * P.nens + e] = (((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]) / (double)((long long)P.nz * ncol);
}
// test hook of the check above: one cell of one variable scaled between the last stage and the final mass (pam_amd_awfl_debug_inject_mass_fault)
__global__ void awfl_poke_kernel(Params P, const double *prim, double *prim_w, double *seed, int ivar, int k, int j, int i, int e, double factor) {
  const long long o = (long long)(k + HS) * P.sz + (long long)j * P.sy + (long long)i * P.sx + e;
  if (ivar < P.nt) seed[(long long)ivar * P.ncell + (((long long)k * P.ny + j) * P.nx + i) * P.nens + e] *= factor;
  else if (ivar == P.nt) { prim_w[P_RHO * P.prim_fs + o] *= factor; prim_w[P_THETA * P.prim_fs + o] /= factor; }   // rho alone: rho*theta unchanged
  else prim_w[P_THETA * P.prim_fs + o] *= factor;
}

// 1 / dz per (level, member) with the reciprocal every kernel forms on the fly (fast_rcp): the z sweep of a folded stage reads it
__global__ void __launch_bounds__(256) awfl_rdz_kernel(const double *__restrict__ dz, double *__restrict__ rdz, long long n) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) rdz[t] = fast_rcp(dz[t]);
}
// mode B of declare_current_profile_as_hydrostatic: the horizontal means of pressure and density
__global__ void __launch_bounds__(64) awfl_hydro_kernel(Params P, const double *__restrict__ prim, double *hy_dens, double *hy_pres) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < (long long)P.nz * P.nens) hydro_cell_mean_body(P, prim, hy_dens, hy_pres, (int)(idx / P.nens), (int)(idx % P.nens));
}

// mode A of declare_current_profile_as_hydrostatic in two steps: the interface pressure of every (face, column) in parallel, then
// the horizontal means in the reference's order (hydro_pint_face, hydro_mean_from_pint).  IDX: 32-bit face index whenever the z
// flux field it borrows has fewer than 2^32 entries (any field that fits the card), 64-bit otherwise.
template <bool VZ_PER_ENS, typename IDX>
__global__ void __launch_bounds__(256) awfl_hydro_pint_kernel(Params P, const double *__restrict__ prim, double *__restrict__ pint) {
  const IDX t = (IDX)blockIdx.x * blockDim.x + threadIdx.x;
  const IDX sz = (IDX)P.sz, sy = (IDX)P.sy, ne = (IDX)P.nens;
  const IDX kf = t / sz, r = t - kf * sz;
  if (kf > (IDX)P.nz) return;
  const IDX j = r / sy, r2 = r - j * sy, i = r2 / ne;
  hydro_pint_face<VZ_PER_ENS>(P, prim, pint, (int)kf, (int)j, (int)i, (int)(r2 - i * ne));
}
__global__ void __launch_bounds__(64) awfl_hydro_sum_kernel(Params P, const double *__restrict__ prim, const double *__restrict__ pint,
                                                            double *grav_var) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < (long long)P.nz * P.nens) hydro_mean_from_pint(P, prim, pint, grav_var, (int)(idx / P.nens), (int)(idx % P.nens));
}

// ------------------------------------------------------------------------------------------------ host side
static thread_local std::string g_last_error;
static int fail(int code, const std::string &msg) {
  g_last_error = msg;
  return code;
}
#define HIP_TRY(expr)                                                                                    \
  do {                                                                                                   \
    hipError_t _e = (expr);                                                                              \
    if (_e != hipSuccess)                                                                                \
      return fail(_e == hipErrorOutOfMemory ? PAM_AMD_ENOMEM : PAM_AMD_ENOGPU,                           \
                  std::string(#expr) + ": " + hipGetErrorString(_e));                                    \
  } while (0)

struct KernelTimer {
  double total_ms = 0;
  long long launches = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

// One ensemble chunk = a contiguous range of members advanced on its own HIP stream, so that the HBM-bound update /
// FCT kernels of one chunk overlap the FP64-VALU-bound flux kernel of another (members are independent; Dycore.h:629-657).
struct Chunk {
  EnsRange r;
  hipStream_t stream = nullptr;   // FCT/update/convert kernels (== the caller's stream when there is a single chunk)
  hipStream_t fstream = nullptr;  // flux kernels (three-kernel stage) / both polynomial kernels (fused stage, chunk 0's is shared):
                                  // a separate HIGHER-priority stream.  The chain of flux kernels is the critical path of a
                                  // three-kernel stage (FP64-issue-bound); with priority its workgroups are replaced as soon
                                  // as they retire and the HBM-bound FCT/update blocks of the previous chunk fill what is left
  hipEvent_t done = nullptr;      // end of this chunk's work in a timeStep (join)
  hipEvent_t flux_done = nullptr; // end of this chunk's most recent flux kernel
  hipEvent_t upd_done = nullptr;  // end of this chunk's most recent update (or init) kernel
};

struct pam_amd_awfl {
  pam_amd_awfl_config_t cfg;
  Params P;
  hipStream_t stream = nullptr;
  int device = 0;               // HIP device the handle was created on (every entry point re-selects it)
  // options (Dycore.h:871-891)
  double R_d, cp_d, R_v, cp_v, p0, grav, cv_d, gamma_d, kappa_d, cv_v, C0;
  // device buffers
  double *prim0 = nullptr, *prim1 = nullptr, *prim2 = nullptr, *flux_x = nullptr, *flux_y = nullptr, *flux_z = nullptr;
  double *seed = nullptr, *mult = nullptr, *dz = nullptr, *grav_var = nullptr, *hy_dens = nullptr, *hy_pres = nullptr;
  double *vz = nullptr, *vert_s2c = nullptr, *vert_wrl = nullptr;
  PowTab *pow_tab = nullptr;    // tables of pow_pos_fast (device)
  // storage the kernels actually use for the dycore's named arrays: the handle's own buffers until the host model binds
  // DataManager-owned storage (pam_amd_awfl_bind_array; the reference's entries are register_and_allocate'd, Dycore.h:868,897-898,983-984)
  double *act_grav_var = nullptr, *act_hy_dens = nullptr, *act_hy_pres = nullptr, *act_vert_s2c = nullptr, *act_vert_wrl = nullptr;
  size_t n_vert_s2c = 0, n_vert_wrl = 0;
  unsigned long long *dt_bits = nullptr;
  unsigned long long *dt_host = nullptr;   // pinned: the CFL minimum read back without blocking the host (time_step)
  unsigned dt_slot = 0;                    // which of the two device words (dt_bits[0 / 1]) the next reduction uses
  unsigned long long *dt_host_dev = nullptr;   // dt_host as the device sees it
  hipEvent_t ev_cfl = nullptr;
  int *fct_flags = nullptr;     // row flags of the FCT multiplier (FctRows in awfl_device.h)
  size_t n_fct_flags = 0, n_fct_lines = 0, n_fct_any = 0;
  int fct_seq = 0;              // launch number of the current stage's FCT kernel: the value a flag must hold to count
  size_t n_prim = 0, n_flux_xy = 0, n_flux_z = 0, n_seed = 0;
  bool timing = false;
  int span_override = 0;       // 0: automatic flux-kernel span
  int chunks_requested = 0;    // 0: automatic
  bool use_priorities = true;  // flux streams get the device's highest stream priority (see Chunk)
  bool interleave_xy = true;
  // timeStep replayed from a captured HIP graph (launch-bound small ensembles; pam_amd_awfl_set_graph_replay)
  struct GraphEntry {
    std::vector<const void *> ptrs;   // the coupler arrays the step reads and writes
    int ncycles; double dt_dyn; const double *prim0_before; long gen;
    hipGraphExec_t exec; double *prim0_after, *prim1_after; int nstages;
  };
  std::vector<GraphEntry> graphs;
  int graph_mode = 0;          // 0 automatic, 1 off, 2 on
  long graph_gen = 0;          // bumped by every setter that changes what a step launches: older graphs are dropped
  bool capturing = false;
  int fail_next_capture = 0;   // test hook: 1 BeginCapture, 2 EndCapture, 3 Instantiate of the NEXT capture reports a failure
  int capture_seq0 = 0;
  int *seq_dev = nullptr;      // device word: fct_seq at the start of the replayed step
  hipStream_t gstream = nullptr;
  hipEvent_t g_fork = nullptr, g_join = nullptr;
  int lane_mode = 0;           // 0 automatic, 1 member lanes, 2 flat (x, member) lanes (pam_amd_awfl_set_lane_mapping)
  bool flat = false;           // resolved lane mapping of the fused stage: flat lanes over (x, member) (small ensembles)
  bool flat_supported = false; // every lane offset fits the 28 bits of the scalar-base + lane-offset addressing
  int xtile_mode = 0;          // 0 automatic, 1 sweep kernels (a wavefront per line span), 2 tile kernels (a lane per cell)
  bool xtile = false;          // resolved: the x direction of the fused stage runs as tile kernels
  int xt_w = 0, xt_tc = 0, xt_lpb = 0;   // tile geometry overrides (0 = automatic)
  int xshuf_mode = 0;          // x tile kernels, exchange between neighbouring cells: 0 automatic, 1 through LDS, 2 wavefront shuffles
  bool xshuf = false;          // resolved: wavefront shuffles (a whole periodic line of a tile lies inside one wavefront)
  // the reference's PAM_DEBUG conservation check as an opt-in (pam_amd_awfl_set_debug_conservation): masses before / after a timeStep
  bool debug_mass = false;
  double *mass_dev = nullptr;  // (2, nt + 2, nens): before, after
  int mass_violations = 0, mass_worst_var = -1, mass_worst_member = -1;
  double mass_max_rel = 0.0;
  std::string mass_report;
  struct { bool armed = false; int ivar, k, j, i, e; double factor; } fault;   // one-shot test hook
  int tail_fuse_mode = 0;      // NT > 1: phase 2 + pressure pass + vapour fix-up as one launch: 0 automatic (on), 1 off, 2 on
  int fold_mode = 0;           // y differences of the state folded into the z sweep's output (P.yz_fold): 0 automatic, 1 off, 2 on
  double *rdz = nullptr;       // (nz, nens) fast_rcp(dz)
  bool independent_ranges = true;    // fused stage, several member ranges: each range's whole stage on its own stream
  int tile_pressure_mode = 0;  // 0 automatic, 1 separate pressure pass, 2 inside the x tile kernel
  bool tile_pressure = true;   // x tile kernels: the next stage's pressure inside awfl_xupd_tile_kernel (no awfl_ptail_kernel launch)
  bool tile_state_parts = false;        // ... and the state pass itself in three parts beside each other (XP_U / XP_VW / XP_T)
  int tile_state_parts_mode = 0;        // 0 automatic, 1 one lane does the whole state pass, 2 three parts
  bool tile_tracers_parallel = false;   // ... and phase 1 of the further tracers in z slices of that launch BESIDE the state pass (idle CUs)
  int ftile_mode = 0;          // 0 automatic, 1 flat-lane sweeps, 2 tile kernel
  bool ftile = true;           // resolved -- flat lanes: the y/z fluxes as ONE tile kernel (a lane per cell) instead of flat-lane sweeps
  int ft_tc_y = 0, ft_tc_z = 0;          // cells / levels per y / z tile (0 = automatic)
  int ft_auto_y = 0, ft_auto_z = 0;      // ... the automatic choice (choose_flux_tiles; 0 = ftile_geometry's own default)
  int ftile_parts_mode = 0;              // 0 automatic, 1 the parts of a tile behind each other (one workgroup), 2 beside each other
  bool ftile_parts = false;              // resolved
  int ncu = 0;                           // compute units of the handle's device
  XTileGeom xg;
  bool fused = false;          // fused x-sweep + state update (needs the third state buffer prim2)
  bool fused_supported = false;
  size_t flux_lds_floor = 0;   // tuning: dynamic LDS requested per flux workgroup (the kernel uses none: a residency cap per CU)
  long long want_units = 3072, two_phase_below = 8192, split_below = 8192;   // launch-shape thresholds (pam_amd_awfl_set_handle_launch_tuning)
  int tracers_per_wave = 0;    // further tracers swept by one wavefront of the separately launched x tracer sweeps: 0 automatic, 1, 2, 4
  bool tracer_prefetch = false;   // phase 2 of those sweeps requests the next trip's loads one trip ahead (pairs only; experiment)
  std::vector<Chunk> chunks;
  hipEvent_t ev_fork = nullptr;
  bool hydro_declared = false;
  std::map<std::string, KernelTimer> timers;
};

namespace {

struct ScopedTimer {
  pam_amd_awfl *h;
  hipStream_t s;
  KernelTimer *t = nullptr;
  hipEvent_t a = nullptr, b = nullptr;
  ScopedTimer(pam_amd_awfl *h_, const char *name, hipStream_t s_) : h(h_), s(s_) {
    if (!h->timing) return;
    t = &h->timers[name];
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipEventRecord(a, s);
  }
  ~ScopedTimer() {
    if (!t) return;
    hipEventRecord(b, s);
    t->pending.emplace_back(a, b);
    t->launches++;
  }
};

void drain(KernelTimer &t) {
  for (auto &p : t.pending) {
    hipEventSynchronize(p.second);
    float ms = 0;
    hipEventElapsedTime(&ms, p.first, p.second);
    t.total_ms += ms;
    hipEventDestroy(p.first);
    hipEventDestroy(p.second);
  }
  t.pending.clear();
}

inline int nblocks(long long n, int bs) { return (int)((n + bs - 1) / bs); }
inline long long ncell_of(const Params &P, EnsRange r) { return (long long)P.nz * P.ny * P.nx * r.ne; }
inline EnsRange full_range(const Params &P) { return EnsRange{0, P.nens}; }

int make_tracer_ptrs(const pam_amd_awfl *h, const pam_amd_awfl_fields_t *f, TracerPtrs &tp) {
  if (!f || !f->density_dry || !f->uvel || !f->vvel || !f->wvel || !f->temp || !f->tracers)
    return fail(PAM_AMD_EINVAL, "fields: null pointer");
  for (int t = 0; t < MAXT; t++) tp.p[t] = nullptr;
  for (int t = 0; t < h->P.nt; t++) {
    if (!f->tracers[t]) return fail(PAM_AMD_EINVAL, "fields: null tracer pointer");
    tp.p[t] = f->tracers[t];
  }
  return PAM_AMD_OK;
}

int launch_init_prim(pam_amd_awfl *h, const pam_amd_awfl_fields_t *f, const pam_amd_awfl_gcm_columns_t *gcm,
                     bool subtract_hy, EnsRange r, hipStream_t s) {
  TracerPtrs tp;
  int rc = make_tracer_ptrs(h, f, tp);
  if (rc) return rc;
  GcmPtrs gp;
  gp.use = gcm ? 1 : 0;
  if (gcm) {
    gp.p[0] = gcm->gcm_density_dry; gp.p[1] = gcm->gcm_temp; gp.p[2] = gcm->gcm_water_vapor;
    gp.p[3] = gcm->gcm_cloud_water; gp.p[4] = gcm->gcm_cloud_ice;
    for (int i = 0; i < 5; i++) if (!gp.p[i]) return fail(PAM_AMD_EINVAL, "gcm columns: null pointer");
  } else {
    for (int i = 0; i < 5; i++) gp.p[i] = nullptr;
  }
  ScopedTimer st(h, "init_prim", s);
  hipLaunchKernelGGL(awfl_init_prim_kernel, cell_grid(h->P, r), dim3(256), 0, s, h->P, r,
                     f->density_dry, f->uvel, f->vvel, f->wvel, f->temp, tp, gp, h->prim0, h->seed, subtract_hy ? 1 : 0);
  HIP_TRY(hipGetLastError());
  return PAM_AMD_OK;
}

int launch_finalize(pam_amd_awfl *h, const pam_amd_awfl_fields_t *f, EnsRange r, hipStream_t s) {
  TracerPtrs tp;
  int rc = make_tracer_ptrs(h, f, tp);
  if (rc) return rc;
  ScopedTimer st(h, "finalize", s);
  hipLaunchKernelGGL(awfl_finalize_kernel, cell_grid(h->P, r), dim3(256), 0, s, h->P, r, h->prim0,
                     h->seed, f->density_dry, f->uvel, f->vvel, f->wvel, f->temp, tp);
  HIP_TRY(hipGetLastError());
  return PAM_AMD_OK;
}

// Span (faces swept by one wavefront) of one sweep: the whole line (up to FLUX_MAX_SPAN faces) if that still leaves enough
// wavefronts to fill the chip (256 CUs x 4 SIMDs x ~4 resident waves, with slack for the tail); otherwise the line is cut,
// down to `min_span` faces (each span re-reads a 5-cell overlap and rebuilds one polynomial).  nlines x ceil(nens/64)
// wavefronts sweep one span each.  `span_override` > 0 forces a value (tests / tuning).
// launch-shape thresholds (wavefronts); settable for experiments through pam_amd_awfl_set_launch_tuning
// The process-wide values below are only the DEFAULTS a new handle starts from (atomics: handles may be created on several host
// threads); every handle carries its own copy (pam_amd_awfl::want_units ...), so tuning one handle never re-shapes another's launches.
static std::atomic<long long> g_want_units{3072};      // a sweep is cut into spans until it has this many wavefronts (or spans reach the shortest)
static std::atomic<long long> g_two_phase_below{8192}; // y/z sweeps: pass 1 and the pairs in launches of their own below this many (line, span) units
static std::atomic<long long> g_split_below{8192};     // x: phase 1 of the further tracers in a launch of its own below this many units
static void choose_span(const pam_amd_awfl *h, int nfaces, long long nlines, int nens, int min_span, int span_override, int &span, int &nspan,
                        int want_scale = 1) {
  const long long nib = nlines * ((nens + 63) / 64);
  // (3072 since round 4, 6144 before: whole x lines and 2 z spans at C2's 128-member shard and at C3 instead of half lines --
  // fewer redundant start-up / closing polynomials; C2@128 +1 ... +3 %, C3 +1.4 %, C4 unchanged: it stops at the shortest span)
  const long long want_units = h->want_units * want_scale;
  if (span_override > 0) {
    span = span_override < FLUX_MAX_SPAN ? span_override : FLUX_MAX_SPAN;
  } else {
    int pieces = (nfaces + FLUX_MAX_SPAN - 1) / FLUX_MAX_SPAN;
    while (nib * pieces < want_units && ((nfaces + pieces - 1) / pieces) > min_span) pieces *= 2;
    span = (nfaces + pieces - 1) / pieces;
  }
  if (span < 1) span = 1;
  nspan = (nfaces + span - 1) / span;
}

// sweeps: bit 0 x, bit 1 y, bit 2 z (the fused stage runs y and z here and x in awfl_xupd_kernel / awfl_xupd_tile_kernel)
int launch_flux(pam_amd_awfl *h, const double *prim, EnsRange r, hipStream_t s, int sweeps = 7, bool diff = false) {
  const Params &P = h->P;
  FluxGrid G;
  if (diff && (sweeps & 1)) return fail(PAM_AMD_EINVAL, "flux launch: the difference form has no x sweep");
  // flat lanes (small ensembles): the y/z sweeps of the fused stage take 64 consecutive items of (level | y, x, member) per
  // wavefront instead of 64 members of one line; one range = the whole ensemble
  const bool flat = h->flat && diff;
  if (flat && (r.e0 != 0 || r.ne != P.nens)) return fail(PAM_AMD_EINVAL, "flux launch: flat lanes sweep the whole ensemble in one range");
  if (flat && h->ftile && sweeps == 6) {
    // tile kernel: a lane per cell, both directions in one launch (small ensembles)
    const FTileGeom Gy = ftile_geometry(P, 1, h->ft_tc_y ? h->ft_tc_y : h->ft_auto_y), Gz = ftile_geometry(P, 2, h->ft_tc_z ? h->ft_tc_z : h->ft_auto_z);
    FTileGroups Q;
    Q.ny_groups = ftile_groups(P, 1, Q.gy, FT_MAXG);
    Q.nz_groups = ftile_groups(P, 2, Q.gz, FT_MAXG);
    const int gyx = Gy.nch * Gy.ntl, nby = P.sim2d ? 0 : gyx * ((P.nz + Gy.lpb - 1) / Gy.lpb), nbz = Gz.nch * Gz.ntl;
    int T = ftile_threads(Gz);
    if (!P.sim2d && ftile_threads(Gy) > T) T = ftile_threads(Gy);
    T = ((T + 63) / 64) * 64;
    if (T > ftile_max_threads(P)) return fail(PAM_AMD_EINVAL, "flux tile launch: a tile must fit a workgroup of 1024 lanes (512 with per-member vertical grids)");
    const size_t lds = (size_t)(2 * FT_NG + 4) * T * sizeof(double);
    ScopedTimer st(h, "flux", s);
    // the parts of a tile (acoustic triple, groups of advected quantities) beside each other in workgroups of their own -- grid.y --
    // while every workgroup still finds a CU of its own (choose: resolve_lane_mapping), else behind each other in one workgroup
    const int maxg = Q.ny_groups > Q.nz_groups ? Q.ny_groups : Q.nz_groups;
    const int parts = h->ftile_parts ? 1 : 0;
    const dim3 fgrid((unsigned)(nby + nbz), parts ? (unsigned)(1 + maxg) : 1u, 1u);
    if (P.vz_per_ens) hipLaunchKernelGGL(awfl_flux_tile_kernel<true>, fgrid, dim3(T), lds, s, P, Gy, Gz, Q, nby, gyx > 0 ? gyx : 1, prim, h->flux_y, h->flux_z, parts);
    else hipLaunchKernelGGL(awfl_flux_tile_kernel<false>, fgrid, dim3(T), lds, s, P, Gy, Gz, Q, nby, gyx > 0 ? gyx : 1, prim, h->flux_y, h->flux_z, parts);
    HIP_TRY(hipGetLastError());
    return PAM_AMD_OK;
  }
  const long long gy = flat ? (flat_items(P, 1) + 63) / 64 : (long long)P.nz * P.nx;   // groups of 64 lanes per member block (member lanes: lines)
  const long long gz = flat ? (flat_items(P, 2) + 63) / 64 : (long long)P.ny * P.nx;
  const int ens_for_span = flat ? 1 : P.nens;
  // the span is chosen from the WHOLE ensemble so that results/scheduling do not depend on the chunking
  choose_span(h, P.nx, (long long)P.nz * P.ny, P.nens, P.seg, h->span_override, G.spx, G.nsx);
  choose_span(h, P.ny, gy, ens_for_span, P.seg, h->span_override, G.spy, G.nsy);
  choose_span(h, P.nz + 1, gz, ens_for_span, P.seg, h->span_override, G.spz, G.nsz);
  if (diff) { G.spy = P.ny; G.nsy = 1; }   // difference form: a periodic line is swept whole (its last cell needs face n == face 0)
  const long long nblk = flat ? 1 : (r.ne + 63) / 64;     // member lanes: a wavefront = 64 consecutive members of ONE line
  const long long nblk_all = flat ? 1 : (P.nens + 63) / 64;
  const long long ux0 = (sweeps & 1) ? (long long)P.nz * P.ny * G.nsx : 0;
  const long long uy0 = (P.sim2d || !(sweeps & 2)) ? 0 : gy * G.nsy;
  const long long uz0 = (sweeps & 4) ? gz * G.nsz : 0;
  // Small ensembles: a wavefront that sweeps its span for pass 1 and then for every pair of advected fields, one after the
  // other, is a long serial chain on a mostly empty chip.  Then pass 1 runs in a launch of its own (`part` 0) and the pairs in a
  // second one with one wavefront per (span, pair) (`part` 1); decided from the WHOLE ensemble (chunking-independent).
  const int npairs = flux_sweep_pairs(P, diff);   // advected fields besides the normal velocity, two per sweep
  // (round 5, measured and removed -- commit 38c84c8, profiles/r05_ab_experiments.txt: the two parts as kernels of their own, each with
  // its own register count -- pass 1 alone 127, pairs 118, one field per wavefront 86 registers -- C4 flux 0.121 -> 0.126 / 0.127 ms,
  // C3 0.411 -> 0.414 / 0.434: the whole-sweep kernel at 128 registers already runs 4 wavefronts per SIMD)
  const bool two_phase = (ux0 + uy0 + uz0) * nblk_all < h->two_phase_below;
  const int nphase = two_phase ? 2 : 1;
  if ((ux0 + uy0 + uz0) * nblk * (two_phase ? npairs : 1) > 0x3fffffffll)
    return fail(PAM_AMD_EINVAL, "flux launch: more than 2^30 wavefronts in one launch");
  // the kernel uses no LDS; a dynamic LDS request only caps its residency per CU when other kernels should co-reside
  size_t lds_bytes = 0;
  if (h->chunks.size() > 1) lds_bytes = h->flux_lds_floor;
  // The z sweep in a launch of its own, BEHIND the x / y sweeps, in two cases:
  //   fold   (fused stage, 3-D, member lanes: P.yz_fold) it reads the y sweep's differences of the state variables and stores the y+z part
  //          of their divergence, one field per variable, which is all the x-sweep then loads (yz_divergence in awfl_device.h);
  //   pe     per-member vertical grids with member lanes: awfl_fluxz_pe_kernel (the levels' tables staged in LDS per workgroup)
  const bool fold = diff && P.yz_fold && !flat && (sweeps & 6) == 6;
  const bool pe = P.vz_per_ens && !flat && (sweeps & 4);
  const int nsub = ((fold || pe) && (sweeps & 3)) ? 2 : 1;
  ScopedTimer st(h, "flux", s);
  for (int sub = 0; sub < nsub; sub++) {
  const int mask = nsub == 1 ? sweeps : (sub == 0 ? (sweeps & 3) : (sweeps & 4));
  const bool zonly = (mask == 4);
  ScopedTimer st2(h, nsub == 1 ? "flux_all" : (sub == 0 ? "flux_xy" : "flux_z"), s);
  for (int phase = 0; phase < nphase; phase++) {
  G.part = two_phase ? phase : -1;
  G.npx = G.npy = G.npz = npairs;
  const long long mult = (G.part == 1) ? npairs : 1;
  G.nux = (mask & 1) ? (int)(ux0 * nblk * mult) : 0; G.nuy = (mask & 2) ? (int)(uy0 * nblk * mult) : 0; G.nuz = (mask & 4) ? (int)(uz0 * nblk * mult) : 0;
  G.nbx = (G.nux + FLUX_WAVES - 1) / FLUX_WAVES; G.nby = (G.nuy + FLUX_WAVES - 1) / FLUX_WAVES;
  G.nbz = (G.nuz + FLUX_WAVES - 1) / FLUX_WAVES;
  G.nbx_l = G.nby_l = 0;
  if (!two_phase && h->interleave_xy && !P.sim2d && G.nux > 0 && G.nuy > 0 && G.nux % (FLUX_WAVES * P.nz) == 0 && G.nuy % (FLUX_WAVES * P.nz) == 0) {
    G.nbx_l = G.nbx / P.nz;
    G.nby_l = G.nby / P.nz;
  }
  if (G.nbx + G.nby + G.nbz == 0) continue;
  if (pe && zonly) {
    // member-block-major workgroups of ZPE_WAVES columns each (awfl_fluxz_pe_kernel)
    const long long ncg = ((long long)P.ny * P.nx + ZPE_WAVES - 1) / ZPE_WAVES;
    const long long nwg = nblk * ncg * G.nsz * mult;
    if (nwg > 0x3fffffffll) return fail(PAM_AMD_EINVAL, "flux launch: more than 2^30 workgroups in one launch");
    const dim3 zgrid((unsigned)nwg), zblock(64 * ZPE_WAVES);
    const bool f = fold && zonly;
#define PAMA_LAUNCH_ZPE(DF, FO)                                                                                         \
  hipLaunchKernelGGL((awfl_fluxz_pe_kernel<DF, FO>), zgrid, zblock, 0, s, P, r, G.spz, G.nsz, G.part, npairs, prim, h->flux_y, h->flux_z)
    if (!diff) PAMA_LAUNCH_ZPE(false, false); else if (f) PAMA_LAUNCH_ZPE(true, true); else PAMA_LAUNCH_ZPE(true, false);
#undef PAMA_LAUNCH_ZPE
    HIP_TRY(hipGetLastError());
    continue;
  }
  const dim3 grid(G.nbx + G.nby + G.nbz), block(FLUX_THREADS);
#define PAMA_LAUNCH_FLUX(VZ, DF, FL, FO)                                                                                \
  hipLaunchKernelGGL((awfl_flux_kernel<VZ, DF, FL, FO>), grid, block, lds_bytes, s, P, G, r, prim, h->flux_x, h->flux_y, h->flux_z)
  if (flat) { if (P.vz_per_ens) PAMA_LAUNCH_FLUX(true, true, true, false); else PAMA_LAUNCH_FLUX(false, true, true, false); }
  else if (fold && zonly) PAMA_LAUNCH_FLUX(false, true, false, true);
  else if (diff) PAMA_LAUNCH_FLUX(false, true, false, false);      // (per-member grids: only the z sweep reads tables, and it is not in this launch)
  else PAMA_LAUNCH_FLUX(false, false, false, false);
#undef PAMA_LAUNCH_FLUX
  HIP_TRY(hipGetLastError());
  }
  }
  return PAM_AMD_OK;
}

// Row flags of the stage that is being launched (FctRows).  next_fct_stage() is called ONCE per tendency stage, before the
// stage's first FCT launch: every ensemble range of the stage then uses the same flag value.
int next_fct_stage(pam_amd_awfl *h) {
  if (h->fct_seq == 0x7fffffff) {   // wrap (once per 2^31 stages): drain everything, forget every flag
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(h->fct_flags, 0, (h->n_fct_flags + h->n_fct_any + h->n_fct_lines) * sizeof(int)));
    h->fct_seq = 0;
  }
  h->fct_seq++;
  return PAM_AMD_OK;
}
// sparse_store: multipliers of unflagged rows are not stored (only the fused stage asks for it: the three-kernel stage
// and the debug entries keep `mult` complete for the tests that read it); needs wavefront == row, i.e. a 64-aligned range
FctRows fct_rows(const pam_amd_awfl *h, EnsRange r, bool sparse_store) {
  FctRows rows;
  rows.flags = h->fct_flags;
  rows.any = h->fct_flags + h->n_fct_flags;                  // one int per member block behind the rows
  rows.lines = h->fct_flags + h->n_fct_flags + h->n_fct_any;   // ... and the line flags behind those
  rows.seq = h->fct_seq;
  rows.seq_base = nullptr;
  if (h->capturing) {        // the launch goes into a graph: the stage number relative to the word the replay sets first
    rows.seq = h->fct_seq - h->capture_seq0;
    rows.seq_base = h->seq_dev;
  }
  rows.sparse_store = (sparse_store && r.e0 % 64 == 0 && r.ne % 64 == 0) ? 1 : 0;
  return rows;
}

// (three-kernel stage only: in the fused stage every tracer's multiplier comes out of its x-sweep)
int launch_fct(pam_amd_awfl *h, double dt, EnsRange r, hipStream_t s) {
  ScopedTimer st(h, "fct_mult", s);
  hipLaunchKernelGGL(awfl_fct_kernel, cell_grid(h->P, r), dim3(256), 0, s, h->P, r, h->flux_x, h->flux_y,
                     h->flux_z, h->seed, h->mult, fct_rows(h, r, false), dt, 0);
  HIP_TRY(hipGetLastError());
  return PAM_AMD_OK;
}

template <int STAGE>
int launch_update(pam_amd_awfl *h, const double *prim_in, const double *prim0, double *prim_out, double dt_dyn, EnsRange r,
                  hipStream_t s) {
  ScopedTimer st(h, "update", s);
  hipLaunchKernelGGL(awfl_update_kernel<STAGE>, cell_grid(h->P, r), dim3(256), 0, s, h->P, r, prim_in,
                     prim0, prim_out, h->flux_x, h->flux_y, h->flux_z, h->mult, fct_rows(h, r, false), h->seed, dt_dyn);
  HIP_TRY(hipGetLastError());
  return PAM_AMD_OK;
}

// Can the last three launches of a stage -- phase 2 of the further tracers, the pressure pass, water vapour's fix-up -- go out as ONE
// (awfl_xtr2_tail_kernel)?  Member-lane sweeps with further tracers, pairs or singles in phase 2 (the measured defaults), and the stage's
// x-sweeps and tail on the same stream.
bool tail_fusable(const pam_amd_awfl *h) {
  const Params &P = h->P;
  if (h->tail_fuse_mode == 1 || h->xtile || P.flat_cells || P.nt < 2) return false;
  if (h->tracers_per_wave == 4 || h->tracer_prefetch) return false;
  return true;
}

template <int STAGE>
int launch_xupd(pam_amd_awfl *h, const double *prim_in, const double *prim0, double *prim_out, double dt_dyn, double dt_stage,
                EnsRange r, hipStream_t s, bool allow_tail_fusion = false, bool *tail_fused = nullptr) {
  const Params &P = h->P;
  if (tail_fused) *tail_fused = false;
  if (h->xtile) {
    // tile kernels: a lane per cell, the whole ensemble in one launch (small ensembles; launches the sweeps cannot fill the chip with)
    if (r.e0 != 0 || r.ne != P.nens) return fail(PAM_AMD_EINVAL, "x-tile launch: the tile kernels take the whole ensemble in one range");
    const XTileGeom &G = h->xg;
    const int nlines = P.nz * P.ny, threads = xtile_threads(G);
    const dim3 block((unsigned)G.W, (unsigned)xtile_rows(G), (unsigned)G.lpb);
    const dim3 grid((unsigned)(G.ntl * G.nmb), (unsigned)((nlines + G.lpb - 1) / G.lpb), 1);
    if (grid.y > 65535u) return fail(PAM_AMD_EINVAL, "x-tile launch: more than 65535 groups of x lines");
    if (!h->xshuf && (size_t)XT_NS * (threads + xtile_stage_elems(G)) * sizeof(double) > 160 * 1024)
      return fail(PAM_AMD_EINVAL, "x-tile launch: the staged tile does not fit the 160 KB of LDS");
    // a wavefront is one row of FCT flags only when a row of the tile is exactly one 64-member block
    const bool wave_is_row = (G.W == 64 && P.nens % 64 == 0);
    // a whole line inside one wavefront: neighbours by wavefront shuffles (no LDS image, no barrier); else through LDS
    const bool shuf = h->xshuf;
    const size_t lds_state = shuf ? 0 : (size_t)XT_NS * (threads + xtile_stage_elems(G)) * sizeof(double);
    const size_t lds_pair = shuf ? 0 : (size_t)2 * (threads + xtile_stage_elems(G)) * sizeof(double);
    {
      ScopedTimer st(h, "xupd", s);
      // phase 1 of the further tracers: 0 in a launch of its own (below), 1 inline behind the state pass, 2 in z slices of this launch
      // beside the state pass (small grids with idle CUs)
      const int npairs_x = (P.nt - 1 + 1) / 2;
      const int sparts = (h->tile_pressure && h->tile_state_parts) ? 1 : 0;      // (the parts exist for the fused form only)
      const int tr_mode = h->tile_pressure ? (((h->tile_tracers_parallel || sparts) && npairs_x > 0 && npairs_x < 65535) ? 2 : 1) : 0;
      const dim3 sgrid(grid.x, grid.y, (unsigned)((sparts ? 3 : 1) + (tr_mode == 2 ? npairs_x : 0)));
      if (shuf)
        hipLaunchKernelGGL((awfl_xupd_tile_kernel<STAGE, true>), sgrid, block, lds_state, s, P, G, prim_in, prim0,
                           prim_out, h->flux_x, h->flux_y, h->flux_z, h->seed, h->mult, fct_rows(h, r, wave_is_row), dt_dyn, dt_stage,
                           h->tile_pressure ? 1 : 0, tr_mode, sparts);
      else
        hipLaunchKernelGGL((awfl_xupd_tile_kernel<STAGE, false>), sgrid, block, lds_state, s, P, G, prim_in, prim0,
                           prim_out, h->flux_x, h->flux_y, h->flux_z, h->seed, h->mult, fct_rows(h, r, wave_is_row), dt_dyn, dt_stage,
                           h->tile_pressure ? 1 : 0, tr_mode, sparts);
      HIP_TRY(hipGetLastError());
    }
    const int npairs = (P.nt - 1 + 1) / 2;
    if (npairs > 65535) return fail(PAM_AMD_EINVAL, "x-tile launch: too many tracer pairs");
    if (npairs > 0) {
      const dim3 tgrid(grid.x, grid.y, (unsigned)npairs);
      if (!h->tile_pressure) {       // (small ensembles: phase 1 ran inline in the state kernel, like the pressure pass)
        ScopedTimer st(h, "xtr1", s);
        if (shuf)
          hipLaunchKernelGGL((awfl_xtr_tile_kernel<STAGE, 1, true>), tgrid, block, lds_pair, s, P, G, prim_in, prim0,
                             prim_out, h->flux_x, h->flux_y, h->flux_z, h->seed, h->mult, fct_rows(h, r, wave_is_row), dt_dyn, dt_stage, 0);
        else
          hipLaunchKernelGGL((awfl_xtr_tile_kernel<STAGE, 1, false>), tgrid, block, lds_pair, s, P, G, prim_in, prim0,
                             prim_out, h->flux_x, h->flux_y, h->flux_z, h->seed, h->mult, fct_rows(h, r, wave_is_row), dt_dyn, dt_stage, 0);
        HIP_TRY(hipGetLastError());
      }
      {
        // (small ensembles: one more z slice does water vapour's fix-up -- launch_tail then has nothing left to launch)
        ScopedTimer st(h, "xtr2", s);
        const dim3 tgrid2(grid.x, grid.y, (unsigned)(npairs + (h->tile_pressure ? 1 : 0)));
        if (shuf)
          hipLaunchKernelGGL((awfl_xtr_tile_kernel<STAGE, 2, true>), tgrid2, block, lds_pair, s, P, G, prim_in, prim0,
                             prim_out, h->flux_x, h->flux_y, h->flux_z, h->seed, h->mult, fct_rows(h, r, wave_is_row), dt_dyn, dt_stage,
                             h->tile_pressure ? 1 : 0);
        else
          hipLaunchKernelGGL((awfl_xtr_tile_kernel<STAGE, 2, false>), tgrid2, block, lds_pair, s, P, G, prim_in, prim0,
                             prim_out, h->flux_x, h->flux_y, h->flux_z, h->seed, h->mult, fct_rows(h, r, wave_is_row), dt_dyn, dt_stage,
                             h->tile_pressure ? 1 : 0);
        HIP_TRY(hipGetLastError());
      }
    }
    return PAM_AMD_OK;
  }
  // wavefronts: (x line, block of 64 members, span of cells).  Normally a wavefront owns a whole line; when the ensemble alone
  // does not fill the chip the lines are cut into spans (each recomputes its closing face) as choose_span decides from the
  // WHOLE ensemble, so that results and schedule do not depend on the chunking
  int span, nspan;
  choose_span(h, P.nx, (long long)P.nz * P.ny, P.nens, P.seg, h->span_override, span, nspan);
  const long long nlb = (long long)P.nz * P.ny * ((r.ne + 63) / 64);
  const long long nunits = nlb * nspan;
  // a wavefront sweeps its cells once for the state and once per pair of further tracers, one after the other: when there
  // are fewer wavefronts than the chip has slots, the tracer sweeps go to their own launch, one wavefront per pair
  const int npairs = (P.nt - 1 + 1) / 2;
  const bool split = npairs > 0 && (long long)P.nz * P.ny * ((P.nens + 63) / 64) * nspan < h->split_below;
  if (nunits * (npairs > 0 ? npairs : 1) > 0x3fffffffll) return fail(PAM_AMD_EINVAL, "x-sweep launch: more than 2^30 wavefronts");
  if (r.e0 % 64) return fail(PAM_AMD_EINVAL, "x-sweep launch: member ranges of the fused stage start at multiples of 64 (a wavefront is one row of FCT flags)");
  {
    ScopedTimer st(h, "xupd", s);
    if (P.yz_fold)      // (the z sweep has left the y+z part of the state's divergence in flux_z: no y differences to load)
      hipLaunchKernelGGL((awfl_xupd_kernel<STAGE, true>), dim3(nblocks(nunits, FLUX_WAVES)), dim3(FLUX_THREADS), 0, s, P, r,
                         prim_in, prim0, prim_out, h->flux_x, h->flux_y, h->flux_z, h->seed, h->mult, fct_rows(h, r, true), dt_dyn,
                         dt_stage, split ? 0 : 1, span, nspan);
    else
      hipLaunchKernelGGL((awfl_xupd_kernel<STAGE, false>), dim3(nblocks(nunits, FLUX_WAVES)), dim3(FLUX_THREADS), 0, s, P, r,
                         prim_in, prim0, prim_out, h->flux_x, h->flux_y, h->flux_z, h->seed, h->mult, fct_rows(h, r, true), dt_dyn,
                         dt_stage, split ? 0 : 1, span, nspan);
    HIP_TRY(hipGetLastError());
  }
  // the tracer launches have npairs wavefronts per (line, member block, span): their lines are cut less (or not at all)
  // Further tracers per wavefront of these launches (0 = automatic; measured on MI355X, round 5, profiles/r05_ab_experiments.txt):
  //   phase 1 -- ONE: 83 instead of 119 registers (5 instead of 4 wavefronts per SIMD), and only the face mass flux is loaded twice:
  //              C4 shard 0.091 -> 0.083 ms, C3 0.206 -> 0.196, C4 whole 0.552 -> 0.530;
  //   phase 2 -- PAIRS (the three densities and the mass flux are shared: singles move 23 % more bytes through a kernel that runs at
  //              5.2 TB/s: C4 shard 0.123 -> 0.142 ms), except for three further tracers, where pairs are one double and one single
  //              wavefront per line and singles three equal ones (C3: 0.316 -> 0.300 ms);
  //   four per wavefront lose everywhere (2 wavefronts per SIMD: C4 shard 0.82 -> 0.75 G although 12 % fewer bytes move).
  const int nfur = P.nt - 1;
  const int per1 = h->tracers_per_wave ? h->tracers_per_wave : 1;
  const int per2 = h->tracers_per_wave ? h->tracers_per_wave : (nfur == 3 ? 1 : 2);
  int tspan = span, tnspan = nspan;
  long long tunits = 0;
  int ngroups = npairs;
  bool quads = false, singles = false;
  auto shape = [&](int per) {        // the launch shape of one phase
    quads = per == 4; singles = per == 1;
    ngroups = quads ? (nfur + 3) / 4 : (singles ? nfur : npairs);     // wavefronts per (line, member block, span)
    tspan = span; tnspan = nspan;
    // (twice the wavefronts of a state sweep before the lines stay whole: the tracer launches are short, register-light kernels of two
    // member ranges that run beside each other -- finer units pack better.  Measured on MI355X (round 5, profiles/r05_ab_experiments.txt),
    // C4 shard, A/B on two boxes: phase 1 in half lines, phase 2 in quarter lines 0.799 -> 0.833 G and 0.822 -> 0.865 G; four times: the same;
    // C3 and C4 whole have enough line blocks either way and keep whole lines)
    if (npairs > 0) choose_span(h, P.nx, (long long)P.nz * P.ny * ngroups, P.nens, P.seg, h->span_override, tspan, tnspan, 2);
    tunits = nlb * tnspan * ngroups;
  };
  if (split) {     // phase 1 of the further tracers (their FCT multipliers) in a launch of its own
    shape(per1);
    ScopedTimer st(h, "xtr1", s);
    if (quads)
      hipLaunchKernelGGL((awfl_xtrn_kernel<STAGE, 1, 4>), dim3(nblocks(tunits, FLUX_WAVES)), dim3(FLUX_THREADS), 0, s, P, r, prim_in,
                         prim0, prim_out, h->flux_x, h->flux_y, h->flux_z, h->seed, h->mult, fct_rows(h, r, true), dt_dyn, dt_stage,
                         ngroups, tspan, tnspan);
    else if (singles)
      hipLaunchKernelGGL((awfl_xtrn_kernel<STAGE, 1, 1>), dim3(nblocks(tunits, FLUX_WAVES)), dim3(FLUX_THREADS), 0, s, P, r, prim_in,
                         prim0, prim_out, h->flux_x, h->flux_y, h->flux_z, h->seed, h->mult, fct_rows(h, r, true), dt_dyn, dt_stage,
                         ngroups, tspan, tnspan);
    else
      hipLaunchKernelGGL((awfl_xtr_kernel<STAGE, 1>), dim3(nblocks(tunits, FLUX_WAVES)), dim3(FLUX_THREADS), 0, s, P, r, prim_in,
                         prim0, prim_out, h->flux_x, h->flux_y, h->flux_z, h->seed, h->mult, fct_rows(h, r, true), dt_dyn, dt_stage,
                         npairs, tspan, tnspan);
    HIP_TRY(hipGetLastError());
  }
  // Phase 2 + the pressure pass + water vapour's fix-up in one launch while the phase-2 launch of this range is less than about two
  // rounds of wavefronts (256 CUs x 4 SIMDs x 3 wavefronts of this kernel): there the two short pointwise launches and their boundaries
  // are ~5 % of the stage and hide inside phase 2's tail; on launches that fill the chip several times over the pressure pass runs
  // faster on its own (8 wavefronts per SIMD instead of 3).  Measured on MI355X (round 6, profiles/r06_ab_experiments.txt): C4 shard
  // 0.891 -> 0.905 G; C4 whole 1.0925 -> 1.075 G and C3, the 3-D four-tracer grid +-0 when forced on.
  bool fuse_tail = false;
  if (npairs > 0 && allow_tail_fusion) {
    shape(per2);
    fuse_tail = h->tail_fuse_mode == 2 || tunits < 6144;
  }
  if (tail_fused) *tail_fused = fuse_tail;
  if (npairs > 0 && fuse_tail) {
    const long long nb_xtr = nblocks(tunits, FLUX_WAVES);
    const dim3 pg = cell_grid(P, r, (P.nz + TAIL_LEVELS - 1) / TAIL_LEVELS);
    const long long nb_pt = (long long)pg.x * pg.y * pg.z;
    const long long nb_fix = nblocks((long long)P.nz * P.ny * ((r.ne + 63) / 64), 4);
    if (nb_xtr + nb_pt + nb_fix > 0x7fffffffll) return fail(PAM_AMD_EINVAL, "tail launch: more than 2^31 workgroups");
    ScopedTimer st(h, "xtr2", s);
    const dim3 grid((unsigned)(nb_xtr + nb_pt + nb_fix));
    if (singles)
      hipLaunchKernelGGL((awfl_xtr2_tail_kernel<STAGE, true>), grid, dim3(FLUX_THREADS), 0, s, P, r, prim_in, prim0, prim_out, h->flux_x,
                         h->flux_y, h->flux_z, h->seed, h->mult, fct_rows(h, r, true), fct_rows(h, r, false), dt_dyn, dt_stage, ngroups, tspan,
                         tnspan, (int)nb_xtr, (int)nb_pt, (int)pg.x);
    else
      hipLaunchKernelGGL((awfl_xtr2_tail_kernel<STAGE, false>), grid, dim3(FLUX_THREADS), 0, s, P, r, prim_in, prim0, prim_out, h->flux_x,
                         h->flux_y, h->flux_z, h->seed, h->mult, fct_rows(h, r, true), fct_rows(h, r, false), dt_dyn, dt_stage, npairs, tspan,
                         tnspan, (int)nb_xtr, (int)nb_pt, (int)pg.x);
    HIP_TRY(hipGetLastError());
  } else if (npairs > 0) {   // phase 2: their complete update, one wavefront per (line, member block, span, pair)
    shape(per2);
    ScopedTimer st(h, "xtr2", s);
    if (quads)
      hipLaunchKernelGGL((awfl_xtrn_kernel<STAGE, 2, 4>), dim3(nblocks(tunits, FLUX_WAVES)), dim3(FLUX_THREADS), 0, s, P, r, prim_in,
                         prim0, prim_out, h->flux_x, h->flux_y, h->flux_z, h->seed, h->mult, fct_rows(h, r, true), dt_dyn, dt_stage,
                         ngroups, tspan, tnspan);
    else if (singles)
      hipLaunchKernelGGL((awfl_xtrn_kernel<STAGE, 2, 1>), dim3(nblocks(tunits, FLUX_WAVES)), dim3(FLUX_THREADS), 0, s, P, r, prim_in,
                         prim0, prim_out, h->flux_x, h->flux_y, h->flux_z, h->seed, h->mult, fct_rows(h, r, true), dt_dyn, dt_stage,
                         ngroups, tspan, tnspan);
    else if (h->tracer_prefetch)      // (experiment (b): phase 2 with the next trip's loads requested one trip ahead)
      hipLaunchKernelGGL((awfl_xtr_kernel<STAGE, 2, true>), dim3(nblocks(tunits, FLUX_WAVES)), dim3(FLUX_THREADS), 0, s, P, r, prim_in,
                         prim0, prim_out, h->flux_x, h->flux_y, h->flux_z, h->seed, h->mult, fct_rows(h, r, true), dt_dyn, dt_stage,
                         npairs, tspan, tnspan);
    else
      hipLaunchKernelGGL((awfl_xtr_kernel<STAGE, 2>), dim3(nblocks(tunits, FLUX_WAVES)), dim3(FLUX_THREADS), 0, s, P, r, prim_in,
                         prim0, prim_out, h->flux_x, h->flux_y, h->flux_z, h->seed, h->mult, fct_rows(h, r, true), dt_dyn, dt_stage,
                         npairs, tspan, tnspan);
    HIP_TRY(hipGetLastError());
  }
  return PAM_AMD_OK;
}

// the pointwise tail of the fused stage: pressure pass, then the tracers' fix-up pass
template <int STAGE>
int launch_tail(pam_amd_awfl *h, const double *prim_in, const double *prim0, double *prim_out, double dt_dyn, EnsRange r,
                hipStream_t s) {
  if (!(h->xtile && h->tile_pressure)) {      // (small ensembles: the x tile kernel has already made the pressure)
    ScopedTimer st(h, "ptail", s);
    const dim3 g = cell_grid(h->P, r, (h->P.nz + TAIL_LEVELS - 1) / TAIL_LEVELS);
    hipLaunchKernelGGL(awfl_ptail_kernel, g, dim3(256), 0, s, h->P, r, prim_out);
    HIP_TRY(hipGetLastError());
  }
  if (!(h->xtile && h->tile_pressure && h->P.nt > 1)) {    // (else: the last slice of the phase-2 tile launch has done it)
    ScopedTimer st(h, "trfix", s);
    if (h->P.flat_cells) {      // small ensembles: a lane per cell
      hipLaunchKernelGGL(awfl_trfix_flat_kernel<STAGE>, cell_grid(h->P, r), dim3(256), 0, s, h->P, prim_in, prim0, prim_out,
                         h->flux_x, h->flux_y, h->flux_z, h->mult, fct_rows(h, r, false), h->seed, dt_dyn);
    } else {
      const long long units = (long long)h->P.nz * h->P.ny * ((r.ne + 63) / 64);   // water vapour only (the others are complete)
      if (units > 0x3fffffffll) return fail(PAM_AMD_EINVAL, "fix-up launch: more than 2^30 wavefronts");
      hipLaunchKernelGGL(awfl_trfix_kernel<STAGE>, dim3(nblocks(units, 4)), dim3(256), 0, s, h->P, r, prim_in, prim0, prim_out,
                         h->flux_x, h->flux_y, h->flux_z, h->mult, fct_rows(h, r, false), h->seed, dt_dyn);
    }
    HIP_TRY(hipGetLastError());
  }
  return PAM_AMD_OK;
}

// Dycore.h:36-58 on the resident state of one member range: slot 0 = before the sub-steps, 1 = after
int launch_mass(pam_amd_awfl *h, int slot, EnsRange r, hipStream_t s) {
  const Params &P = h->P;
  hipLaunchKernelGGL(awfl_mass_kernel, dim3((unsigned)((r.ne + 63) / 64), (unsigned)(P.nt + 2)), dim3(256), 0, s, P, r, h->prim0, h->seed,
                     h->mass_dev + (size_t)slot * (P.nt + 2) * P.nens);
  HIP_TRY(hipGetLastError());
  return PAM_AMD_OK;
}
// Dycore.h:224-251: a (variable, member) pair is reported when its mass changed by more than 1e-10 relative AND absolute
int evaluate_mass(pam_amd_awfl *h) {
  const Params &P = h->P;
  const size_t n = (size_t)(P.nt + 2) * P.nens;
  std::vector<double> m(2 * n);
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(m.data(), h->mass_dev, 2 * n * sizeof(double), hipMemcpyDeviceToHost));
  h->mass_violations = 0; h->mass_max_rel = 0.0; h->mass_worst_var = h->mass_worst_member = -1;
  h->mass_report.clear();
  for (int ivar = 0; ivar < P.nt + 2; ivar++)
    for (int e = 0; e < P.nens; e++) {
      const double a = m[(size_t)ivar * P.nens + e], b = m[n + (size_t)ivar * P.nens + e];
      const double ad = std::fabs(b - a), rd = ad / (std::fabs(a) + 1.e-20);
      const bool bad = !(rd <= 1.e-10) && !(ad <= 1.e-10);      // (a NaN mass is a violation)
      if (rd > h->mass_max_rel || (rd != rd && h->mass_worst_var < 0)) { h->mass_max_rel = rd; h->mass_worst_var = ivar; h->mass_worst_member = e; }
      if (!bad) continue;
      h->mass_violations++;
      if (h->mass_violations <= 32) {
        char line[256];
        std::snprintf(line, sizeof(line), "WARNING: conservation violated variable,ensemble,rel_diff,mass_diff,init,final: %d , %d , %.6e , %.6e , %.6e , %.6e\n",
                      ivar, e, rd, b - a, a, b);
        h->mass_report += line;
      }
    }
  return PAM_AMD_OK;
}

// The CFL minimum over this handle's members (Dycore.h:86-101) in two halves: the reduction, which leaves its result in pinned host
// memory, is queued on the caller's stream (launch); the host waits for that read-back alone (finish) -- time_step queues the conversion of
// the coupler state in between, so the device works while the host wakes up and issues the stages.
int launch_cfl(pam_amd_awfl *h, const pam_amd_awfl_fields_t *f, double cfl, bool lean) {
  if (!f || !f->tracers) return fail(PAM_AMD_EINVAL, "fields: null pointer");
  int nb = nblocks(h->P.ncell, 256);
  if (nb > 2048) nb = 2048;
  if (lean) {
    // two device words, used in turn: both start from +inf (create), and every reduction puts +inf back into the word of the next one,
    // whose last user has been read by then (finish_cfl)
    unsigned long long *word = h->dt_bits + (h->dt_slot & 1u), *next_word = h->dt_bits + ((h->dt_slot + 1u) & 1u);
    h->dt_slot++;
    ScopedTimer st(h, "cfl", h->stream);
    hipLaunchKernelGGL(awfl_cfl_kernel<true>, dim3(nb), dim3(256), 0, h->stream, h->P, f->density_dry, f->uvel, f->vvel, f->wvel,
                       f->temp, f->tracers[h->P.idWV], cfl, word, next_word, (unsigned *)(h->dt_bits + 2), h->dt_host_dev);
    HIP_TRY(hipGetLastError());
  } else {
    unsigned long long *word = h->dt_bits + 3;      // (a word of its own: the two forms may alternate on one handle)
    HIP_TRY(hipMemcpyAsync(word, h->dt_host + 1, sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));   // +inf, pinned
    {
      ScopedTimer st(h, "cfl", h->stream);
      hipLaunchKernelGGL(awfl_cfl_kernel<false>, dim3(nb), dim3(256), 0, h->stream, h->P, f->density_dry, f->uvel, f->vvel, f->wvel,
                         f->temp, f->tracers[h->P.idWV], cfl, word, (unsigned long long *)nullptr, (unsigned *)nullptr,
                         (unsigned long long *)nullptr);
      HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemcpyAsync(h->dt_host, word, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
  }
  HIP_TRY(hipEventRecord(h->ev_cfl, h->stream));
  return PAM_AMD_OK;
}
int finish_cfl(pam_amd_awfl *h, double *dt) {
  HIP_TRY(hipEventSynchronize(h->ev_cfl));
  double v;
  std::memcpy(&v, h->dt_host, sizeof(v));
  *dt = v;
  return PAM_AMD_OK;
}
int local_time_step(pam_amd_awfl *h, const pam_amd_awfl_fields_t *f, double cfl, double *dt) {
  if (int rc = launch_cfl(h, f, cfl, h->chunks.size() <= 1)) return rc;
  return finish_cfl(h, dt);
}

// The internal streams of the handles of a process are SHARED: the k-th member range of every handle on a device uses the same stream
// (one per device, kind and priority), created on first use and never destroyed.  The runtime maps streams onto a handful of hardware
// queues in the order of their creation, and the two member ranges of a handle only overlap when their streams sit on different queues:
// a process that had created a second set of streams -- another handle alive at the same time, or a handle re-chunked -- ran every
// later two-range workload in a degraded state for the rest of its life (one GPU's shard of C4 0.90 -> 0.60 G, C2 at 128 members
// 2.45 -> 2.17 G: tools/probe/two_handles.py; the "slow mode" of bench.py's other configurations in round 6).  Sharing is safe: every
// time_step forks its ranges from the caller's stream with an event and joins them back into it, so handles that share a range stream
// are simply ordered on it in the order of their calls.
struct SharedStream { hipStream_t s; int device, priority, kind, ordinal; };
static std::mutex g_stream_pool_mutex;
static std::vector<SharedStream> g_stream_pool;
static hipError_t shared_stream(int device, int priority, int kind, int ordinal, hipStream_t *out) {
  std::lock_guard<std::mutex> lk(g_stream_pool_mutex);
  for (auto &p : g_stream_pool)
    if (p.device == device && p.priority == priority && p.kind == kind && p.ordinal == ordinal) { *out = p.s; return hipSuccess; }
  hipStream_t s = nullptr;
  const hipError_t e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority);
  if (e != hipSuccess) return e;
  g_stream_pool.push_back({s, device, priority, kind, ordinal});
  *out = s;
  return hipSuccess;
}

void destroy_chunks(pam_amd_awfl *h) {
  for (auto &c : h->chunks) {
    if (c.done) (void)hipEventDestroy(c.done);
    if (c.flux_done) (void)hipEventDestroy(c.flux_done);
    if (c.upd_done) (void)hipEventDestroy(c.upd_done);
    // (the range streams are shared and stay; what this handle queued on them has been joined into the caller's stream by time_step)
  }
  h->chunks.clear();
}

// every captured step is dropped when something changes what a step launches (a setter, a re-bound array)
void drop_graphs(pam_amd_awfl *h) {
  // (a replay may still be queued on gstream: an exec is destroyed only once nothing can be running it -- ADVICE r4)
  if (!h->graphs.empty() && h->gstream) {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur != h->device) (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->gstream);
    if (cur >= 0 && cur != h->device) (void)hipSetDevice(cur);
  }
  for (auto &g : h->graphs) (void)hipGraphExecDestroy(g.exec);
  h->graphs.clear();
  h->graph_gen++;
}

// Tile sizes of the y/z flux TILE kernel (small grids: below 2.6e5 cells).  One launch, one workgroup size T = the larger of the y and
// the z tile: such a launch is a latency problem -- it is over when the most loaded CU is -- so (cells per y tile, levels per z tile)
// are chosen to minimise (workgroups per CU, rounded up) x T, among geometries whose launched lanes are at least 80 % active; ties go
// to fewer launched lanes.  Measured on MI355X (round 5, profiles/r05_ab_experiments.txt), 32x32x60 with one member: 340 workgroups of
// 512 lanes (11, 6) -> 248 of 640 (16, 8): 17.8 -> 16.1 us, 0.645 -> 0.688 G; a geometry with 276 workgroups of 768: 22.2 us; the
// reference's 250 x 1 x 50 shape: 36 workgroups of 512 -> 100 of 256: 12.8 -> 11.3 us.
void choose_flux_tiles(pam_amd_awfl *h) {
  const Params &P = h->P;
  h->ft_auto_y = h->ft_auto_z = 0;
  if (h->ncu <= 0 || P.ncell > 262144) return;
  long long best_cost = -1, best_lanes = 0;
  for (int ty = 0; ty <= (P.sim2d ? 0 : (P.ny < 30 ? P.ny : 30)); ty++) {         // (0: ftile_geometry's own choice, e.g. whole short lines)
    if (ty == 1) continue;
    for (int tz = 2; tz <= 14; tz++) {
      const FTileGeom Gy = ftile_geometry(P, 1, ty), Gz = ftile_geometry(P, 2, tz);
      if (ty > 0 && Gy.tc != ty) continue;                          // (clamped: the same geometry as a smaller request)
      if (Gz.tc != tz) continue;
      const long long nby = P.sim2d ? 0 : (long long)Gy.nch * Gy.ntl * ((P.nz + Gy.lpb - 1) / Gy.lpb), nbz = (long long)Gz.nch * Gz.ntl;
      int T = ftile_threads(Gz);
      if (!P.sim2d && ftile_threads(Gy) > T) T = ftile_threads(Gy);
      T = ((T + 63) / 64) * 64;
      if (T > ftile_max_threads(P)) continue;
      const long long launched = (nby + nbz) * T;
      const long long active = (P.sim2d ? 0 : nby * ftile_threads(Gy)) + nbz * ftile_threads(Gz);
      if (active * 5 < launched * 4) continue;
      const long long cost = ((nby + nbz + h->ncu - 1) / h->ncu) * T;
      if (best_cost < 0 || cost < best_cost || (cost == best_cost && launched < best_lanes)) {
        best_cost = cost; best_lanes = launched; h->ft_auto_y = P.sim2d ? 0 : ty; h->ft_auto_z = tz;
      }
    }
  }
}

// The fold (P.yz_fold) is OFF unless asked for: measured on MI355X (round 6, profiles/r06_ab_experiments.txt, A/B on one box) it moves
// four loads per cell from the HBM-bound x-sweep (3.59 -> 3.31 ms at C2) into the FP64-issue-bound flux kernel, which pays the same
// back (4.36 -> 4.59 ms for the y and z launches together): C2 2.555 -> 2.564 G, 256 members 2.548 -> 2.537, 128 members 2.466 ->
// 2.445, the 3-D four-tracer grid at 256 members 1.416 -> 1.381.
#ifndef PAMA_FOLD_DEFAULT
#define PAMA_FOLD_DEFAULT 0
#endif
// Lane mapping of the fused stage (decided from the WHOLE ensemble; results never depend on it):
//   flat   the y/z sweeps take 64 consecutive (x, member) items per wavefront (flat_lane) instead of 64 members of one line
//   xtile  the x direction runs as tile kernels (a lane per cell) instead of sweeps (a wavefront per line span)
//   P.flat_cells  the pointwise kernels run on a grid that is flat over every cell
// Automatic: all three for ensembles of fewer than 64 members (a member-lane wavefront would be mostly idle lanes).
void resolve_lane_mapping(pam_amd_awfl *h) {
  Params &P = h->P;
  drop_graphs(h);
  const bool small = P.nens < 64;
  h->flat_supported = P.prim_fs < (1ll << 28) && P.fz_fs < (1ll << 28);
  // flat y/z lanes also for RAGGED ensembles below 128 members (70 members in member lanes are two wavefronts per line, the second
  // one with 6 lanes: measured 1.53 -> 1.71 G at 70, 1.99 -> 2.09 G at 96 on 32x32x60); from 128 on the two independent member
  // ranges, which need member lanes, are worth more
  const bool ragged = P.nens % 64 != 0 && P.nens < 128;
  h->flat = h->flat_supported && (h->lane_mode == 2 || (h->lane_mode == 0 && (small || ragged)));
  h->xg = xtile_geometry(P, h->xt_w, h->xt_tc, h->xt_lpb, h->ncu);
  // (the groups of x lines are the y dimension of the tile kernels' launch grid: at most 65535)
  const bool grid_ok = ((long long)P.nz * P.ny + h->xg.lpb - 1) / h->xg.lpb <= 65535;
  h->xtile = xtile_supported(P) && grid_ok && (h->xtile_mode == 2 || (h->xtile_mode == 0 && small));
  // neighbours by wavefront shuffles instead of an LDS image + barriers wherever a line lies inside one wavefront
  h->xshuf = h->xtile && h->xshuf_mode != 1 && xtile_line_in_wavefront(P, h->xg);
  P.flat_cells = (h->lane_mode != 1 && (long long)P.nx * P.nens < 256 && P.ncell < (1ll << 31)) ? 1 : 0;
  // the y differences of the state folded into what the z sweep stores (3-D, member-lane sweeps in y, z AND x: the tile kernels form
  // the y+z part of the divergence themselves, with the same yz_divergence -- same bits either way)
  P.yz_fold = (!P.sim2d && !h->flat && !h->xtile && (h->fold_mode == 2 || (h->fold_mode == 0 && PAMA_FOLD_DEFAULT))) ? 1 : 0;
  // the y/z fluxes of a flat-lane stage: ONE tile kernel (a lane per cell) while the whole ensemble is below ~2.6e5 cells -- a flat-lane
  // sweep is then a handful of wavefronts walking their lines serially -- and flat-lane sweeps above (they read every input once and
  // build no halo rows; measured on MI355X, 32x32x60: 1 member 67 -> 20 us per stage, 8 members 85 -> 82, 32 members 184 -> 320)
  h->ftile = h->ftile_mode == 2 || (h->ftile_mode == 0 && P.ncell <= 262144);
  choose_flux_tiles(h);
  // the parts of a flux tile BESIDE each other while every part's workgroup can be resident at once (16 wavefronts per CU at the
  // kernel's ~110 registers); measured (round 5): the 250 x 1 x 50 shape, 100 -> 300 workgroups of 256 lanes: 12.7 -> 9.6 us; 32x32x60 with
  // one member, 248 -> 744 workgroups of 640 lanes (one per CU at a time): 16.8 -> 24.4 us
  {
    const FTileGeom Gy = ftile_geometry(P, 1, h->ft_tc_y ? h->ft_tc_y : h->ft_auto_y), Gz = ftile_geometry(P, 2, h->ft_tc_z ? h->ft_tc_z : h->ft_auto_z);
    const long long nby = P.sim2d ? 0 : (long long)Gy.nch * Gy.ntl * ((P.nz + Gy.lpb - 1) / Gy.lpb), nbz = (long long)Gz.nch * Gz.ntl;
    int T = ftile_threads(Gz);
    if (!P.sim2d && ftile_threads(Gy) > T) T = ftile_threads(Gy);
    T = ((T + 63) / 64) * 64;
    const int nadv = 3 + P.nt - (P.sim2d ? 1 : 0), maxg = (nadv + FT_NG - 1) / FT_NG;      // groups of advected quantities per sweep
    const long long per_cu = T > 0 ? 16 / (T / 64) : 0;
    h->ftile_parts = h->ftile_parts_mode == 2 ||
                     (h->ftile_parts_mode == 0 && h->ncu > 0 && per_cu > 0 && (nby + nbz) * (1 + maxg) <= (long long)h->ncu * per_cu);
  }
  // the pressure pass inside the x tile kernel while a stage is a handful of short launches (one launch of ~10 us less); above, the
  // separate pass with the pow tables in LDS and 6 levels per lane is cheaper than the tile kernel's longer lanes
  h->tile_pressure = h->tile_pressure_mode == 2 || h->tile_pressure_mode == 3 || (h->tile_pressure_mode == 0 && P.ncell <= 1048576);
  // phase 1 of the further tracers beside the state pass instead of behind it, while the launch's workgroups fit the chip about twice
  // over (16 wavefronts per CU at the kernel's ~120 registers; the tracer workgroups rebuild the face mass flux: two more polynomials
  // per cell on SIMDs that would be idle).  Measured (round 5, profiles/r05_ab_experiments.txt): the 250 x 1 x 50 shape with 4 tracers,
  // 50 -> 150 workgroups: x kernel 18.5 -> 13.6 us, 0.112 -> 0.127 G; 32x32x60 with 4 tracers: one member 0.450 -> 0.496 G, two 0.651 ->
  // 0.688 G; a single 2-D column set with 10 tracers (32 x 1 x 60): x kernel 20.7 -> 11.6 us
  {
    const long long nwg = (long long)h->xg.ntl * h->xg.nmb * (((long long)P.nz * P.ny + h->xg.lpb - 1) / h->xg.lpb);
    const int npairs_x = (P.nt - 1 + 1) / 2;
    const int T = ((xtile_threads(h->xg) + 63) / 64) * 64;
    const long long per_cu = 16 / (T / 64 > 0 ? T / 64 : 1);       // workgroups a CU holds at the kernel's ~120 registers (16 wavefronts)
    // the state pass itself in three parts beside each other (they rebuild the face mass flux and the new density) only while every
    // workgroup of the launch finds a CU of its OWN: the polynomials are a minority of a lane's chain (two dependent rounds of loads
    // and the launch itself are the rest), so the parts buy little -- measured (round 5): 250 x 1 x 50 with 4 tracers, 150 -> 250
    // workgroups: x kernel 13.9 -> 12.8 us (0.135 -> 0.139 G); 32x32x60 with one member, 240 -> 720 workgroups: 14.6 -> 14.8 us; two: 17.5 -> 20.0
    h->tile_state_parts = h->tile_pressure && (h->tile_state_parts_mode == 2 ||
                          (h->tile_state_parts_mode == 0 && h->ncu > 0 && nwg * (3 + npairs_x) <= (long long)h->ncu));
    h->tile_tracers_parallel = h->tile_pressure && npairs_x > 0 &&
                               (h->tile_pressure_mode == 3 ||
                                (h->tile_pressure_mode == 0 && h->ncu > 0 && per_cu > 0 && nwg * (1 + npairs_x) <= 2 * (long long)h->ncu * per_cu));
  }
}

// (Re)build the chunk list: n contiguous member ranges whose sizes are multiples of 64 where possible.
int build_chunks(pam_amd_awfl *h) {
  (void)hipStreamSynchronize(h->stream);
  if (h->gstream) (void)hipStreamSynchronize(h->gstream);
  drop_graphs(h);
  for (auto &c : h->chunks) {
    if (c.stream && c.stream != h->stream) (void)hipStreamSynchronize(c.stream);
    if (c.fstream && c.fstream != h->stream) (void)hipStreamSynchronize(c.fstream);
  }
  destroy_chunks(h);
  const int nens = h->P.nens;
  int n = h->chunks_requested;
  if (n <= 0) {
    // automatic.  Fused stage: ONE range -- its two big kernels are each bound by their own resource (flux y,z: the FP64
    // pipe; fused x-sweep: HBM) and lose more from sharing the chip than the pointwise tail gains (measured on C2, round 2:
    // 1/2/8 ranges -> 2.08/2.05/1.72 G cell-updates/s).  Three-kernel stage: ranges on internal streams so that the HBM-bound
    // update of one range runs beside the FP64-bound flux kernel of the next; chunking only pays when every range's flux
    // launch still fills the chip (W = wavefronts of one whole-ensemble flux launch).
    const Params &P = h->P;
    int sp, nsx, nsy, nsz;
    choose_span(h, P.nx, (long long)P.nz * P.ny, P.nens, P.seg, h->span_override, sp, nsx);
    choose_span(h, P.ny, (long long)P.nz * P.nx, P.nens, P.seg, h->span_override, sp, nsy);
    choose_span(h, P.nz + 1, (long long)P.ny * P.nx, P.nens, P.seg, h->span_override, sp, nsz);
    const long long nblk = (P.nens + 63) / 64;
    const long long ux = (long long)P.nz * P.ny * nblk * nsx, uy = (long long)P.nz * P.nx * nblk * nsy;
    const long long uz = (long long)P.ny * P.nx * nblk * nsz;
    const long long W = ux + (P.sim2d ? 0 : uy) + uz;
    // ~11.8 k wave-units of flux work per chunk (128 members of a 32x32x60 CRM) measured best on MI355X for 256..2048
    // members; smaller jobs run as one chunk
    n = (int)((W + 8000) / 11776);
    if (n < 1) n = 1;
    if (n > 16) n = 16;
    // Fused stage: TWO ranges, each running its whole stage on its own stream (independent_ranges below): the launches of a stage
    // overlap their ramp-up and drain phases with the other range's kernels.  Measured on MI355X (round 4, 1 -> 2 independent ranges):
    // C2 at 128 members 2.29 -> 2.37 G, at 256 2.44 -> 2.52, at 1024 2.600 -> 2.610; C3 1.84 -> 1.93; C4 0.79 -> 0.81; four
    // ranges lose (C4 0.70, C3 1.83), and so do two ranges that share one compute stream (round 2's schedule: C4 0.65).
    if (h->fused) n = (nens >= 128) ? 2 : 1;
  }
  if (h->fused && (h->flat || h->xtile)) n = 1;    // flat lanes and tile kernels take the whole ensemble in one launch
  if (h->P.flat_cells) n = 1;                      // (pointwise kernels with a flat grid over every cell)
  const int per = (((nens + n - 1) / n + 63) / 64) * 64;
  for (int e0 = 0; e0 < nens; e0 += per) {
    Chunk c;
    c.r.e0 = e0;
    c.r.ne = (e0 + per <= nens) ? per : nens - e0;
    h->chunks.push_back(c);
  }
  if (h->chunks.size() == 1) {
    h->chunks[0].stream = h->stream;
    h->chunks[0].fstream = h->stream;
  } else {
    int prio_low = 0, prio_high = 0;   // numerically lower = higher priority
    HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
    // (streams are a scarce resource: the runtime maps them onto a handful of hardware queues, and two "independent" ranges whose
    // streams land on one queue run one after the other -- seen in round 4: one more internal stream per handle cost C4 0.80 -> 0.54 G.
    // Independent ranges of the fused stage need ONE stream each; only the shared-compute-stream schedules use a second one.)
    const bool one_stream_per_range = h->fused && h->independent_ranges;
    int ordinal = 0;
    for (auto &c : h->chunks) {
      HIP_TRY(shared_stream(h->device, prio_low, 0, ordinal, &c.stream));
      if (one_stream_per_range) c.fstream = c.stream;
      else HIP_TRY(shared_stream(h->device, h->use_priorities ? prio_high : prio_low, 1, ordinal, &c.fstream));
      ordinal++;
      HIP_TRY(hipEventCreateWithFlags(&c.done, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&c.flux_done, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&c.upd_done, hipEventDisableTiming));
    }
  }
  return PAM_AMD_OK;
}

void free_all(pam_amd_awfl *h) {
  drop_graphs(h);
  if (h->gstream) { (void)hipStreamSynchronize(h->gstream); (void)hipStreamDestroy(h->gstream); h->gstream = nullptr; }
  if (h->g_fork) (void)hipEventDestroy(h->g_fork);
  if (h->g_join) (void)hipEventDestroy(h->g_join);
  h->g_fork = h->g_join = nullptr;
  if (h->seq_dev) (void)hipFree(h->seq_dev);
  h->seq_dev = nullptr;
  destroy_chunks(h);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  h->ev_fork = nullptr;
  double **bufs[] = {&h->prim0, &h->prim1, &h->prim2, &h->flux_x, &h->flux_y, &h->flux_z, &h->seed, &h->mult, &h->dz,
                     &h->grav_var, &h->hy_dens, &h->hy_pres, &h->vz, &h->vert_s2c, &h->vert_wrl, &h->rdz};
  for (auto b : bufs) {
    if (*b) (void)hipFree(*b);
    *b = nullptr;
  }
  if (h->dt_bits) (void)hipFree(h->dt_bits);
  h->dt_bits = nullptr;
  if (h->dt_host) (void)hipHostFree(h->dt_host);
  h->dt_host = nullptr;
  if (h->ev_cfl) (void)hipEventDestroy(h->ev_cfl);
  h->ev_cfl = nullptr;
  if (h->mass_dev) (void)hipFree(h->mass_dev);
  h->mass_dev = nullptr;
  if (h->pow_tab) (void)hipFree(h->pow_tab);
  h->pow_tab = nullptr;
  if (h->fct_flags) (void)hipFree(h->fct_flags);
  h->fct_flags = nullptr;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ C ABI
// every entry point that touches the GPU runs on the handle's device, whatever the caller's current device is
#define USE_DEVICE(h)                                                                  \
  do {                                                                                 \
    int _cur = -1;                                                                     \
    if (hipGetDevice(&_cur) != hipSuccess || _cur != (h)->device) HIP_TRY(hipSetDevice((h)->device)); \
  } while (0)

// shared with the other translation units of the library (modules_kernels.hip)
extern "C" int pam_amd_set_last_error_(int code, const char *msg) { return fail(code, msg ? msg : ""); }

extern "C" {

int pam_amd_awfl_abi_version(void) { return PAM_AMD_AWFL_ABI_VERSION; }
const char *pam_amd_awfl_last_error(void) { return g_last_error.c_str(); }

int pam_amd_awfl_init(const pam_amd_awfl_config_t *cfg, pam_amd_awfl_t **out) {
  if (!cfg || !out) return fail(PAM_AMD_EINVAL, "init: null argument");
  *out = nullptr;
  if (cfg->nens < 1 || cfg->nx < 3 || cfg->nz < 3 || cfg->ny < 1 || (cfg->ny > 1 && cfg->ny < 3))
    return fail(PAM_AMD_EINVAL, "init: need nens>=1, nx>=3, nz>=3 and ny==1 or ny>=3 (periodic stencils of half-width 3)");
  if (cfg->num_tracers < 1 || cfg->num_tracers > MAXT)
    return fail(PAM_AMD_EINVAL, "init: num_tracers must be in [1,50] (pam_const.h max_fields)");
  if (cfg->idWV < 0 || cfg->idWV >= cfg->num_tracers) return fail(PAM_AMD_EINVAL, "init: idWV out of range (tracer water_vapor missing?)");
  if (!(cfg->xlen > 0) || !(cfg->ylen > 0)) return fail(PAM_AMD_EINVAL, "init: xlen/ylen must be positive (set_grid not called?)");
  if (!cfg->tracer_positive || !cfg->tracer_adds_mass || !cfg->vertical_cell_dz)
    return fail(PAM_AMD_EINVAL, "init: null tracer flags or vertical_cell_dz");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail(PAM_AMD_ENOGPU, "init: no HIP device available (this library has no CPU path)");

  pam_amd_awfl *h = new pam_amd_awfl();
  h->cfg = *cfg;
  if (hipGetDevice(&h->device) != hipSuccess) { delete h; return fail(PAM_AMD_ENOGPU, "init: hipGetDevice failed"); }
  if (hipDeviceGetAttribute(&h->ncu, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || h->ncu < 1) h->ncu = 256;
  h->stream = (hipStream_t)cfg->stream;
  auto opt = [](double v, double dflt) { return std::isnan(v) ? dflt : v; };
  h->R_d = opt(cfg->R_d, 287.);   h->cp_d = opt(cfg->cp_d, 1003.);
  h->R_v = opt(cfg->R_v, 461.);   h->cp_v = opt(cfg->cp_v, 1859.);
  h->p0 = opt(cfg->p0, 1.e5);     h->grav = opt(cfg->grav, 9.81);
  // Dycore.h:877-890: each derived constant is set only if the option is absent, then all are read back from the coupler
  // (:942-950) -- a host model that pre-set one of them gets ITS value (NaN = absent = derive)
  h->cv_d = opt(cfg->cv_d, h->cp_d - h->R_d);
  h->gamma_d = opt(cfg->gamma_d, h->cp_d / h->cv_d);
  h->kappa_d = opt(cfg->kappa_d, h->R_d / h->cp_d);
  h->cv_v = opt(cfg->cv_v, h->R_v - h->cp_v);
  h->C0 = opt(cfg->C0, std::pow(h->R_d * std::pow(h->p0, -h->kappa_d), h->gamma_d));

  Params &P = h->P;
  std::memset(&P, 0, sizeof(P));
  P.nens = cfg->nens; P.nx = cfg->nx; P.ny = cfg->ny; P.nz = cfg->nz; P.nt = cfg->num_tracers;
  P.sim2d = (cfg->ny == 1);
  P.grav_balance = 1;   // Dycore.h:866
  P.seg = 8;
  P.dx = cfg->xlen / cfg->nx; P.dy = cfg->ylen / cfg->ny; P.rdx = 1.0 / P.dx; P.rdy = 1.0 / P.dy;
  P.C0 = h->C0; P.gamma = h->gamma_d; P.grav = h->grav; P.R_d = h->R_d; P.R_v = h->R_v;
  P.sx = P.nens; P.sy = (long long)P.nx * P.nens; P.sz = (long long)P.ny * P.nx * P.nens;
  P.prim_fs = (long long)(P.nz + 2 * HS) * P.sz;
  P.ncell = (long long)P.nz * P.sz;
  P.fz_fs = (long long)(P.nz + 1) * P.sz;
  P.idWV = cfg->idWV;
  for (int t = 0; t < P.nt; t++) {
    if (cfg->tracer_positive[t]) P.pos_mask |= (1ull << t);
    if (cfg->tracer_adds_mass[t]) P.mass_mask |= (1ull << t);
  }

  const size_t nzn = (size_t)P.nz * P.nens;
  std::vector<double> dz_host(nzn);
  auto bail = [&](int code, const std::string &m) { free_all(h); delete h; return fail(code, m); };
#define INIT_TRY(expr)                                                                                    \
  do {                                                                                                    \
    hipError_t _e = (expr);                                                                               \
    if (_e != hipSuccess)                                                                                 \
      return bail(_e == hipErrorOutOfMemory ? PAM_AMD_ENOMEM : PAM_AMD_ENOGPU, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)
  INIT_TRY(hipMemcpy(dz_host.data(), cfg->vertical_cell_dz, nzn * sizeof(double), hipMemcpyDefault));
  for (size_t i = 0; i < nzn; i++)
    if (!(dz_host[i] > 0)) return bail(PAM_AMD_EINVAL, "init: vertical_cell_dz must be positive (set_grid not called?)");
  VerticalTables vt = build_vertical_tables(dz_host.data(), P.nz, P.nens);
  P.vz_per_ens = vt.per_ens ? 1 : 0;

  h->n_prim = (size_t)(6 + P.nt) * P.prim_fs;
  h->n_flux_xy = (size_t)(5 + P.nt) * P.ncell;
  h->n_flux_z = (size_t)(5 + P.nt) * P.fz_fs;
  h->n_seed = (size_t)P.nt * P.ncell;
  INIT_TRY(hipMalloc(&h->prim0, h->n_prim * 8));
  INIT_TRY(hipMalloc(&h->prim1, h->n_prim * 8));
  // the fused x-sweep writes into a third buffer (nothing may be updated in place along a periodic line)
  h->fused_supported = true;
  h->fused = h->fused_supported;
  if (h->fused_supported) INIT_TRY(hipMalloc(&h->prim2, h->n_prim * 8));
  INIT_TRY(hipMalloc(&h->flux_x, h->n_flux_xy * 8));
  INIT_TRY(hipMalloc(&h->flux_y, (P.sim2d ? 8 : h->n_flux_xy) * 8));
  INIT_TRY(hipMalloc(&h->flux_z, h->n_flux_z * 8));
  INIT_TRY(hipMalloc(&h->seed, h->n_seed * 8));
  INIT_TRY(hipMalloc(&h->mult, h->n_seed * 8));
  h->n_fct_flags = (size_t)P.nt * (size_t)P.nz * P.ny * P.nx * (size_t)((P.nens + 63) / 64);   // FctRows: (nt, nz, ny, nx, blocks of 64 members)
  h->n_fct_lines = (size_t)P.nt * (size_t)P.nz * P.ny * (size_t)((P.nens + 63) / 64);          // (nt, nz, ny, blocks): line flags
  h->n_fct_any = (size_t)((P.nens + 63) / 64);                                                  // "some row of this member block"
  INIT_TRY(hipMalloc(&h->fct_flags, (h->n_fct_flags + h->n_fct_any + h->n_fct_lines) * sizeof(int)));   // rows, the "any row flagged in this stage" word, lines
  INIT_TRY(hipMemset(h->fct_flags, 0, (h->n_fct_flags + h->n_fct_any + h->n_fct_lines) * sizeof(int)));
  h->fct_seq = 0;
  INIT_TRY(hipMalloc(&h->dz, nzn * 8));
  INIT_TRY(hipMalloc(&h->grav_var, nzn * 8));
  INIT_TRY(hipMalloc(&h->hy_dens, nzn * 8));
  INIT_TRY(hipMalloc(&h->hy_pres, nzn * 8));
  INIT_TRY(hipMalloc(&h->vz, vt.table.size() * 8));
  INIT_TRY(hipMalloc(&h->vert_s2c, vt.s2c.size() * 8));
  INIT_TRY(hipMalloc(&h->vert_wrl, vt.wrl.size() * 8));
  // CFL minimum: [0], [1] the copy-free form's two words (used in turn), [2] its count of finished workgroups, [3] the other form's word;
  // on the host (pinned): [0] the minimum as read, [1] +inf
  INIT_TRY(hipMalloc(&h->dt_bits, 32));
  INIT_TRY(hipHostMalloc((void **)&h->dt_host, 16, hipHostMallocDefault));
  h->dt_host[0] = 0;
  h->dt_host[1] = 0x7FF0000000000000ull;
  INIT_TRY(hipHostGetDevicePointer((void **)&h->dt_host_dev, h->dt_host, 0));
  {
    const unsigned long long w4[4] = {0x7FF0000000000000ull, 0x7FF0000000000000ull, 0ull, 0x7FF0000000000000ull};
    INIT_TRY(hipMemcpy(h->dt_bits, w4, sizeof(w4), hipMemcpyHostToDevice));
  }
  INIT_TRY(hipEventCreateWithFlags(&h->ev_cfl, hipEventDisableTiming));
  {
    PowTab pt;
    build_pow_tab(pt);
    INIT_TRY(hipMalloc(&h->pow_tab, sizeof(PowTab)));
    INIT_TRY(hipMemcpy(h->pow_tab, &pt, sizeof(PowTab), hipMemcpyHostToDevice));
    P.pw = h->pow_tab;
  }
  INIT_TRY(hipMemcpy(h->dz, dz_host.data(), nzn * 8, hipMemcpyHostToDevice));
  INIT_TRY(hipMalloc(&h->rdz, nzn * 8));
  hipLaunchKernelGGL(awfl_rdz_kernel, dim3((unsigned)((nzn + 255) / 256)), dim3(256), 0, h->stream, h->dz, h->rdz, (long long)nzn);
  INIT_TRY(hipGetLastError());
  P.rdz = h->rdz;
  INIT_TRY(hipMemcpy(h->vz, vt.table.data(), vt.table.size() * 8, hipMemcpyHostToDevice));
  INIT_TRY(hipMemcpy(h->vert_s2c, vt.s2c.data(), vt.s2c.size() * 8, hipMemcpyHostToDevice));
  INIT_TRY(hipMemcpy(h->vert_wrl, vt.wrl.data(), vt.wrl.size() * 8, hipMemcpyHostToDevice));
  // the reference leaves variable_gravity / hy_* unset until declare_current_profile_as_hydrostatic;
  // poison them so a missing call is loud (NaN state) instead of silently wrong.
  INIT_TRY(hipMemset(h->grav_var, 0xFF, nzn * 8));
  INIT_TRY(hipMemset(h->hy_dens, 0xFF, nzn * 8));
  INIT_TRY(hipMemset(h->hy_pres, 0xFF, nzn * 8));
  P.dz = h->dz; P.grav_var = h->grav_var; P.hy_dens = h->hy_dens; P.hy_pres = h->hy_pres; P.vz = h->vz;
  h->act_grav_var = h->grav_var; h->act_hy_dens = h->hy_dens; h->act_hy_pres = h->hy_pres;
  h->act_vert_s2c = h->vert_s2c; h->act_vert_wrl = h->vert_wrl;
  h->n_vert_s2c = vt.s2c.size(); h->n_vert_wrl = vt.wrl.size();
  INIT_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
  // (the graph-replay stream, its events and the stage-number word are created when the replay is switched on)
  // the flux kernel may request more than the default 64 KiB of dynamic LDS (residency cap)
  INIT_TRY(hipFuncSetAttribute((const void *)awfl_flux_kernel<false, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  INIT_TRY(hipFuncSetAttribute((const void *)awfl_flux_kernel<false, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  INIT_TRY(hipFuncSetAttribute((const void *)awfl_flux_kernel<false, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  // the tile kernels stage their tiles in LDS: up to 14 doubles per lane of a 1024-lane workgroup
  INIT_TRY(hipFuncSetAttribute((const void *)awfl_xupd_tile_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  INIT_TRY(hipFuncSetAttribute((const void *)awfl_xupd_tile_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  INIT_TRY(hipFuncSetAttribute((const void *)awfl_xupd_tile_kernel<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  INIT_TRY(hipFuncSetAttribute((const void *)awfl_xtr_tile_kernel<1, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  INIT_TRY(hipFuncSetAttribute((const void *)awfl_xtr_tile_kernel<1, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  INIT_TRY(hipFuncSetAttribute((const void *)awfl_xtr_tile_kernel<2, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  INIT_TRY(hipFuncSetAttribute((const void *)awfl_xtr_tile_kernel<2, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  INIT_TRY(hipFuncSetAttribute((const void *)awfl_xtr_tile_kernel<3, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  INIT_TRY(hipFuncSetAttribute((const void *)awfl_xtr_tile_kernel<3, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  INIT_TRY(hipFuncSetAttribute((const void *)awfl_flux_tile_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  INIT_TRY(hipFuncSetAttribute((const void *)awfl_flux_tile_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#undef INIT_TRY
  h->flux_lds_floor = 0;   // no residency cap by default
  h->want_units = g_want_units.load(); h->two_phase_below = g_two_phase_below.load(); h->split_below = g_split_below.load();
  resolve_lane_mapping(h);
  if (int rc = build_chunks(h)) { free_all(h); delete h; return rc; }
  *out = h;
  return PAM_AMD_OK;
}

int pam_amd_awfl_finalize(pam_amd_awfl_t *h) {
  if (!h) return PAM_AMD_OK;
  USE_DEVICE(h);
  (void)hipStreamSynchronize(h->stream);
  if (h->gstream) (void)hipStreamSynchronize(h->gstream);
  for (auto &c : h->chunks) {
    if (c.stream && c.stream != h->stream) (void)hipStreamSynchronize(c.stream);
    if (c.fstream && c.fstream != h->stream) (void)hipStreamSynchronize(c.fstream);
  }
  for (auto &kv : h->timers) drain(kv.second);
  free_all(h);
  delete h;
  return PAM_AMD_OK;
}

const char *pam_amd_awfl_dycore_name(const pam_amd_awfl_t *) { return "SSPRK3+WENO+FV A-grid (MI355X native)"; }

int pam_amd_awfl_get_option(const pam_amd_awfl_t *h, const char *key, double *value) {
  if (!h || !key || !value) return fail(PAM_AMD_EINVAL, "get_option: null argument");
  const std::string k(key);
  if (k == "R_d") *value = h->R_d;
  else if (k == "R_v") *value = h->R_v;
  else if (k == "cp_d") *value = h->cp_d;
  else if (k == "cp_v") *value = h->cp_v;
  else if (k == "p0") *value = h->p0;
  else if (k == "grav") *value = h->grav;
  else if (k == "cv_d") *value = h->cv_d;
  else if (k == "cv_v") *value = h->cv_v;
  else if (k == "gamma_d") *value = h->gamma_d;
  else if (k == "kappa_d") *value = h->kappa_d;
  else if (k == "C0") *value = h->C0;
  else if (k == "balance_hydrostasis_with_gravity") *value = h->P.grav_balance;
  else if (k == "idWV") *value = h->P.idWV;
  else return fail(PAM_AMD_EINVAL, "ERROR: option " + k + " does not exist");   // Options.h get_option -> endrun
  return PAM_AMD_OK;
}

int pam_amd_awfl_set_balance_hydrostasis_with_gravity(pam_amd_awfl_t *h, int value) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  drop_graphs(h);
  h->P.grav_balance = value ? 1 : 0;
  h->hydro_declared = false;
  return PAM_AMD_OK;
}

int pam_amd_awfl_get_array(pam_amd_awfl_t *h, const char *name, double **device_ptr, int dims[5], int *ndims) {
  if (!h || !name || !device_ptr || !dims || !ndims) return fail(PAM_AMD_EINVAL, "get_array: null argument");
  const std::string k(name);
  const Params &P = h->P;
  if (k == "variable_gravity" || k == "hy_dens_cells" || k == "hy_pressure_cells") {
    *device_ptr = (k == "variable_gravity") ? h->act_grav_var : (k == "hy_dens_cells" ? h->act_hy_dens : h->act_hy_pres);
    dims[0] = P.nz; dims[1] = P.nens; *ndims = 2;
  } else if (k == "vert_sten_to_coefs") {
    *device_ptr = h->act_vert_s2c; dims[0] = P.nz + 2; dims[1] = 5; dims[2] = 5; dims[3] = P.nens; *ndims = 4;
  } else if (k == "vert_weno_recon_lower") {
    *device_ptr = h->act_vert_wrl; dims[0] = P.nz + 2; dims[1] = 3; dims[2] = 3; dims[3] = 3; dims[4] = P.nens; *ndims = 5;
  } else {
    return fail(PAM_AMD_EINVAL, "ERROR: array " + k + " is not owned by the dycore");
  }
  return PAM_AMD_OK;
}

int pam_amd_awfl_bind_array(pam_amd_awfl_t *h, const char *name, double *device_ptr) {
  if (!h || !name || !device_ptr) return fail(PAM_AMD_EINVAL, "bind_array: null argument");
  USE_DEVICE(h);
  const std::string k(name);
  const size_t nzn = (size_t)h->P.nz * h->P.nens;
  double **act = nullptr;
  size_t n = 0;
  if (k == "variable_gravity") { act = &h->act_grav_var; n = nzn; }
  else if (k == "hy_dens_cells") { act = &h->act_hy_dens; n = nzn; }
  else if (k == "hy_pressure_cells") { act = &h->act_hy_pres; n = nzn; }
  else if (k == "vert_sten_to_coefs") { act = &h->act_vert_s2c; n = h->n_vert_s2c; }
  else if (k == "vert_weno_recon_lower") { act = &h->act_vert_wrl; n = h->n_vert_wrl; }
  else return fail(PAM_AMD_EINVAL, "ERROR: array " + k + " is not owned by the dycore");
  if (*act != device_ptr) {   // carry the current contents over, then switch
    HIP_TRY(hipMemcpyAsync(device_ptr, *act, n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    *act = device_ptr;
  }
  drop_graphs(h);
  h->P.grav_var = h->act_grav_var; h->P.hy_dens = h->act_hy_dens; h->P.hy_pres = h->act_hy_pres;
  return PAM_AMD_OK;
}

int pam_amd_awfl_declare_current_profile_as_hydrostatic(pam_amd_awfl_t *h, const pam_amd_awfl_fields_t *fields,
                                                        const pam_amd_awfl_gcm_columns_t *gcm) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  USE_DEVICE(h);
  int rc = launch_init_prim(h, fields, gcm, /*subtract_hy=*/false, full_range(h->P), h->stream);
  if (rc) return rc;
  {
    ScopedTimer st(h, "hydro", h->stream);
    const long long n = (long long)h->P.nz * h->P.nens;
    if (h->P.grav_balance) {
      // mode A: the interface pressures in parallel into the (idle) z flux array, then the means in the reference's order
      const long long nf = h->P.fz_fs;       // (nz + 1) faces x columns
      const dim3 grid(nblocks(nf, 256)), block(256);
      const bool wide = nf >= (1ll << 32);
      if (h->P.vz_per_ens && wide) hipLaunchKernelGGL((awfl_hydro_pint_kernel<true, unsigned long long>), grid, block, 0, h->stream, h->P, h->prim0, h->flux_z);
      else if (h->P.vz_per_ens) hipLaunchKernelGGL((awfl_hydro_pint_kernel<true, unsigned>), grid, block, 0, h->stream, h->P, h->prim0, h->flux_z);
      else if (wide) hipLaunchKernelGGL((awfl_hydro_pint_kernel<false, unsigned long long>), grid, block, 0, h->stream, h->P, h->prim0, h->flux_z);
      else hipLaunchKernelGGL((awfl_hydro_pint_kernel<false, unsigned>), grid, block, 0, h->stream, h->P, h->prim0, h->flux_z);
      hipLaunchKernelGGL(awfl_hydro_sum_kernel, dim3(nblocks(n, 64)), dim3(64), 0, h->stream, h->P, h->prim0, h->flux_z, h->act_grav_var);
    } else
      hipLaunchKernelGGL(awfl_hydro_kernel, dim3(nblocks(n, 64)), dim3(64), 0, h->stream, h->P, h->prim0, h->act_hy_dens, h->act_hy_pres);
    HIP_TRY(hipGetLastError());
  }
  h->hydro_declared = true;
  return PAM_AMD_OK;
}

int pam_amd_awfl_compute_time_step(pam_amd_awfl_t *h, const pam_amd_awfl_fields_t *fields, double cfl, double *dt) {
  if (!h || !dt) return fail(PAM_AMD_EINVAL, "compute_time_step: null argument");
  USE_DEVICE(h);
  return local_time_step(h, fields, cfl, dt);
}

int pam_amd_awfl_convert_coupler_to_dynamics(pam_amd_awfl_t *h, const pam_amd_awfl_fields_t *fields) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  USE_DEVICE(h);
  return launch_init_prim(h, fields, nullptr, !h->P.grav_balance, full_range(h->P), h->stream);
}

int pam_amd_awfl_convert_dynamics_to_coupler(pam_amd_awfl_t *h, const pam_amd_awfl_fields_t *fields) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  USE_DEVICE(h);
  return launch_finalize(h, fields, full_range(h->P), h->stream);
}

int pam_amd_awfl_convert_coupler_to_dynamics_arrays(pam_amd_awfl_t *h, const pam_amd_awfl_fields_t *fields, double *state,
                                                    double *tracers) {
  if (!h || !state || !tracers) return fail(PAM_AMD_EINVAL, "convert_coupler_to_dynamics: null argument");
  USE_DEVICE(h);
  TracerPtrs tp;
  if (int rc = make_tracer_ptrs(h, fields, tp)) return rc;
  const EnsRange r = full_range(h->P);
  hipLaunchKernelGGL(awfl_c2d_arrays_kernel, cell_grid(h->P, r), dim3(256), 0, h->stream, h->P, r, fields->density_dry, fields->uvel,
                     fields->vvel, fields->wvel, fields->temp, tp, state, tracers);
  HIP_TRY(hipGetLastError());
  return PAM_AMD_OK;
}

int pam_amd_awfl_convert_dynamics_to_coupler_arrays(pam_amd_awfl_t *h, const pam_amd_awfl_fields_t *fields, const double *state,
                                                    const double *tracers) {
  if (!h || !state || !tracers) return fail(PAM_AMD_EINVAL, "convert_dynamics_to_coupler: null argument");
  USE_DEVICE(h);
  TracerPtrs tp;
  if (int rc = make_tracer_ptrs(h, fields, tp)) return rc;
  const EnsRange r = full_range(h->P);
  hipLaunchKernelGGL(awfl_d2c_arrays_kernel, cell_grid(h->P, r), dim3(256), 0, h->stream, h->P, r, state, tracers, fields->density_dry,
                     fields->uvel, fields->vvel, fields->wvel, fields->temp, tp);
  HIP_TRY(hipGetLastError());
  return PAM_AMD_OK;
}

int pam_amd_awfl_time_step(pam_amd_awfl_t *h, const pam_amd_awfl_fields_t *fields, double crm_dt, double dt_dyn_hint,
                           int *ncycles_out, double *dt_dyn_out) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  if (!(crm_dt > 0)) return fail(PAM_AMD_EINVAL, "time_step: option crm_dt must be positive");
  USE_DEVICE(h);
  if (!h->hydro_declared)
    return fail(PAM_AMD_ESTATE, "time_step: declare_current_profile_as_hydrostatic has not been called since init / "
                                "since the balance option changed (variable_gravity / hy_* are undefined, SURVEY F4)");
  TracerPtrs tp_check;
  int rc = make_tracer_ptrs(h, fields, tp_check);
  if (rc) return rc;
  // Dycore.h:141-145: the dynamics step comes from the COUPLER fields of every member (before any conversion)
  double dt_dyn = dt_dyn_hint;
  int ncycles = 0;
  // the number of cycles and their length from the dynamics step (Dycore.h:142-145)
  auto settle_cycles = [&]() -> int {
    if (!(dt_dyn > 0) || !std::isfinite(dt_dyn))
      return fail(PAM_AMD_EINVAL, "time_step: CFL time step is not positive/finite (NaN or non-physical coupler state)");
    ncycles = (int)std::ceil(crm_dt / dt_dyn);
    dt_dyn = crm_dt / ncycles;
    if (ncycles_out) *ncycles_out = ncycles;
    if (dt_dyn_out) *dt_dyn_out = dt_dyn;
    return PAM_AMD_OK;
  };
  // Without a hint the step comes from a reduction and a read-back.  A step of ONE member range issued eagerly queues the reduction,
  // then the conversion coupler -> dycore (which does not depend on it), and only then waits for the 8 bytes: the device converts
  // while the host wakes up and issues the stages (A/B on one box: 32x32x60 with one member 0.669 -> 0.687 G, the reference's
  // 250x1x50 input shape 0.1405 -> 0.1525 G).  Not with several ranges: their conversions then start together instead of one host
  // launch apart and the ranges run in lock-step (one GPU's shard of C4: 0.894 -> 0.883 G; started one conversion apart: 0.871).
  // A replayed graph needs the cycle count before it is chosen: there the wait comes first, too.
  const bool graph_on = h->graph_mode == 2;
  bool dt_pending = false;
  if (!(dt_dyn > 0)) {
    if ((rc = launch_cfl(h, fields, 0.8, h->chunks.size() <= 1))) return rc;
    if (graph_on || h->chunks.size() > 1) {
      if ((rc = finish_cfl(h, &dt_dyn))) return rc;
    } else {
      dt_pending = true;
    }
  }
  if (!dt_pending && (rc = settle_cycles())) return rc;
  const bool forked = h->chunks.size() > 1;
  if (forked) {   // chunk streams start after everything already queued on the caller's stream
    HIP_TRY(hipEventRecord(h->ev_fork, h->stream));
    for (auto &c : h->chunks) HIP_TRY(hipStreamWaitEvent(c.stream, h->ev_fork, 0));
  }
  // Everything between fork and join runs in a lambda: whatever fails in there (a launch, an event call), the join below
  // still happens, so that the caller's stream stays ordered after the work already queued on the internal streams.
  auto enqueue = [&]() -> int {
    int rc = PAM_AMD_OK;
    // Dycore.h:128-134 (per chunk)
    for (auto &c : h->chunks) {
      if ((rc = launch_init_prim(h, fields, nullptr, !h->P.grav_balance, c.r, c.stream))) return rc;
      if (h->debug_mass && (rc = launch_mass(h, 0, c.r, c.stream))) return rc;      // Dycore.h:136-138 (after the clipping of :130-134)
      if (forked) HIP_TRY(hipEventRecord(c.upd_done, c.stream));
    }
    if (dt_pending) {
      dt_pending = false;
      if ((rc = finish_cfl(h, &dt_dyn))) return rc;
      if ((rc = settle_cycles())) return rc;
    }
    // Launches are issued stage by stage, round-robin over the chunks.  With several chunks:
    //  * flux kernels run on each chunk's high-priority flux stream and are chained ACROSS chunks by events
    //    (A1 -> B1 -> C1 -> A2 ...): two VALU-bound flux kernels never share the chip;
    //  * a chunk's HBM-bound FCT/update kernels run on its normal-priority stream beside the NEXT chunk's flux kernel.
    hipEvent_t prev_flux = nullptr;
    // One tendency stage of one chunk.  Unfused: flux (x,y,z) -> FCT -> update.  Fused: flux (y,z) -> x-sweeps + update of the
    // state and of every tracer -> pressure pass + tracer fix-up pass (pout differs from pin and p0).
    auto stage = [&](Chunk &c, int st, const double *pin, const double *p0, double *pout, double dt_stage) -> int {
      int r2;
      if (h->fused) {
        // Fused stage.  The two polynomial kernels (flux y,z and the fused x-sweep) of ALL chunks run back to back on ONE
        // high-priority compute stream, chunk after chunk; a chunk's tail (pressure pass, tracer fix-up)
        // runs on the chunk's own stream beside the NEXT chunk's flux kernel.  The x-sweep is both VALU- and HBM-heavy and
        // gets the chip to itself.
        // (independent_ranges: every range runs its whole stage on its own stream -- no shared compute stream, no events between
        // ranges: launches that do not fill the chip overlap their ramp-up and drain phases with another range's kernels)
        const bool indep = forked && h->independent_ranges;
        hipStream_t cs = indep ? c.stream : (forked ? h->chunks[0].fstream : c.stream);
        if (forked && !indep) HIP_TRY(hipStreamWaitEvent(cs, c.upd_done, 0));         // this chunk's previous tail / init
        if ((r2 = launch_flux(h, pin, c.r, cs, 6, true))) return r2;
        const bool allow = (cs == c.stream) && tail_fusable(h);      // (the tail as part of the phase-2 launch)
        bool fuse_tail = false;
        if (st == 1) r2 = launch_xupd<1>(h, pin, p0, pout, dt_dyn, dt_stage, c.r, cs, allow, &fuse_tail);
        else if (st == 2) r2 = launch_xupd<2>(h, pin, p0, pout, dt_dyn, dt_stage, c.r, cs, allow, &fuse_tail);
        else r2 = launch_xupd<3>(h, pin, p0, pout, dt_dyn, dt_stage, c.r, cs, allow, &fuse_tail);
        if (r2) return r2;
        if (forked && !indep) {
          HIP_TRY(hipEventRecord(c.flux_done, cs));
          HIP_TRY(hipStreamWaitEvent(c.stream, c.flux_done, 0));
        }
        if (!fuse_tail) {
          if (st == 1) r2 = launch_tail<1>(h, pin, p0, pout, dt_dyn, c.r, c.stream);
          else if (st == 2) r2 = launch_tail<2>(h, pin, p0, pout, dt_dyn, c.r, c.stream);
          else r2 = launch_tail<3>(h, pin, p0, pout, dt_dyn, c.r, c.stream);
          if (r2) return r2;
        }
        if (forked) HIP_TRY(hipEventRecord(c.upd_done, c.stream));
        return PAM_AMD_OK;
      }
      if (forked) {
        HIP_TRY(hipStreamWaitEvent(c.fstream, c.upd_done, 0));               // this chunk's previous update / init
        if (prev_flux) HIP_TRY(hipStreamWaitEvent(c.fstream, prev_flux, 0));  // the flux kernel launched just before
      }
      if ((r2 = launch_flux(h, pin, c.r, c.fstream, 7))) return r2;
      if (forked) {
        HIP_TRY(hipEventRecord(c.flux_done, c.fstream));
        prev_flux = c.flux_done;
        HIP_TRY(hipStreamWaitEvent(c.stream, c.flux_done, 0));
      }
      if ((r2 = launch_fct(h, dt_stage, c.r, c.stream))) return r2;
      if (st == 1) r2 = launch_update<1>(h, pin, p0, pout, dt_dyn, c.r, c.stream);
      else if (st == 2) r2 = launch_update<2>(h, pin, p0, pout, dt_dyn, c.r, c.stream);
      else r2 = launch_update<3>(h, pin, p0, pout, dt_dyn, c.r, c.stream);
      if (r2) return r2;
      if (forked) HIP_TRY(hipEventRecord(c.upd_done, c.stream));
      return PAM_AMD_OK;
    };
    for (int ic = 0; ic < ncycles; ic++) {
      double *A = h->prim0, *B = h->prim1, *C = h->prim2;
      if (h->fused) {
        // three rotating buffers: A (sub-step start) -> B -> C -> B; the new state B becomes prim0
        if ((rc = next_fct_stage(h))) return rc;
        for (auto &c : h->chunks)   // stage 1 (Dycore.h:156-176)
          if ((rc = stage(c, 1, A, A, B, dt_dyn))) return rc;
        if ((rc = next_fct_stage(h))) return rc;
        for (auto &c : h->chunks)   // stage 2 (Dycore.h:180-200)
          if ((rc = stage(c, 2, B, A, C, (1.0 / 4.0) * dt_dyn))) return rc;
        if ((rc = next_fct_stage(h))) return rc;
        for (auto &c : h->chunks)   // stage 3 (Dycore.h:204-221)
          if ((rc = stage(c, 3, C, A, B, (2.0 / 3.0) * dt_dyn))) return rc;
        h->prim0 = B;
        h->prim1 = A;
      } else {
        // pointwise update kernel: stage 2 and 3 update in place
        if ((rc = next_fct_stage(h))) return rc;
        for (auto &c : h->chunks)
          if ((rc = stage(c, 1, A, A, B, dt_dyn))) return rc;
        if ((rc = next_fct_stage(h))) return rc;
        for (auto &c : h->chunks)
          if ((rc = stage(c, 2, B, A, B, (1.0 / 4.0) * dt_dyn))) return rc;
        if ((rc = next_fct_stage(h))) return rc;
        for (auto &c : h->chunks)
          if ((rc = stage(c, 3, B, A, A, (2.0 / 3.0) * dt_dyn))) return rc;
      }
    }
    // Dycore.h:224-251 (PAM_DEBUG), then :254 (per chunk)
    for (auto &c : h->chunks) {
      if (h->debug_mass) {
        if (h->fault.armed && h->fault.e >= c.r.e0 && h->fault.e < c.r.e0 + c.r.ne) {
          hipLaunchKernelGGL(awfl_poke_kernel, dim3(1), dim3(1), 0, c.stream, h->P, h->prim0, h->prim0, h->seed, h->fault.ivar, h->fault.k,
                             h->fault.j, h->fault.i, h->fault.e, h->fault.factor);
          HIP_TRY(hipGetLastError());
          h->fault.armed = false;
        }
        if ((rc = launch_mass(h, 1, c.r, c.stream))) return rc;
      }
      if ((rc = launch_finalize(h, fields, c.r, c.stream))) return rc;
    }
    return PAM_AMD_OK;
  };
  // Launch-bound ensembles: the whole step (coupler -> dycore, 3 x ncycles stages, dycore -> coupler) is captured ONCE into a HIP
  // graph on an internal stream and replayed; the only thing that changes from one step to the next -- the stage number the FCT
  // flags are compared with -- is a device word set in front of every replay (FctRows::seq_base).
  // OFF unless asked for: measured on MI355X / ROCm 7.2 (round 4, 20 steps each) a replayed step is SLOWER than the same launches
  // issued eagerly -- 32x32x60 with one member 0.907 -> 1.011 ms, two 1.134 -> 1.249, eight 3.41 -> 3.58, the 250x1x50 shape 0.364 ->
  // 0.425 ms: the gaps between DEPENDENT kernels (~2.3 us each) are the same inside a graph, and the fork / join events cost more
  // than the host-side launch calls they replace (the host is not the bottleneck: launches are issued well ahead of the device).
  if (graph_on && h->fused && !forked && !h->timing && !h->debug_mass && h->gstream) {
    const int nstages = 3 * ncycles;
    if (h->fct_seq > 0x7fffffff - nstages - 4) {      // the wrap of the stage number cannot happen inside a graph
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipMemset(h->fct_flags, 0, (h->n_fct_flags + h->n_fct_any + h->n_fct_lines) * sizeof(int)));
      h->fct_seq = 0;
    }
    std::vector<const void *> ptrs = {fields->density_dry, fields->uvel, fields->vvel, fields->wvel, fields->temp};
    for (int t = 0; t < h->P.nt; t++) ptrs.push_back(fields->tracers[t]);
    pam_amd_awfl::GraphEntry *entry = nullptr;
    for (auto &g : h->graphs)
      if (g.gen == h->graph_gen && g.ncycles == ncycles && g.dt_dyn == dt_dyn && g.prim0_before == h->prim0 && g.ptrs == ptrs) entry = &g;
    HIP_TRY(hipEventRecord(h->g_fork, h->stream));
    HIP_TRY(hipStreamWaitEvent(h->gstream, h->g_fork, 0));
    int seq0 = h->fct_seq;
    if (!entry) {
      if (h->graphs.size() >= 8) {                      // (a few shapes of a step recur: both buffer parities x the cycle counts met)
        HIP_TRY(hipStreamSynchronize(h->gstream));      // an earlier replay may still be running one of them
        for (auto &g : h->graphs) (void)hipGraphExecDestroy(g.exec);
        h->graphs.clear();
      }
      pam_amd_awfl::GraphEntry e;
      e.ptrs = ptrs; e.ncycles = ncycles; e.dt_dyn = dt_dyn; e.prim0_before = h->prim0; e.gen = h->graph_gen; e.nstages = nstages;
      Chunk &c0 = h->chunks[0];
      const hipStream_t s_keep = c0.stream, f_keep = c0.fstream;
      c0.stream = c0.fstream = h->gstream;
      h->capturing = true;
      h->capture_seq0 = h->fct_seq;
      // what the capture advances on the host side (stage number, buffer rotation) is put back when no graph comes out of it: no
      // kernel has run then, and the next step must start from the same buffers
      const int seq_keep = h->fct_seq;
      double *const p0_keep = h->prim0, *const p1_keep = h->prim1;
      // (test hook: pam_amd_awfl_debug_fail_next_capture makes one of the three capture calls "fail")
      const int inject = h->fail_next_capture;
      h->fail_next_capture = 0;
      hipError_t cerr = (inject == 1) ? hipErrorUnknown : hipStreamBeginCapture(h->gstream, hipStreamCaptureModeThreadLocal);
      if (cerr == hipSuccess) rc = enqueue();
      hipGraph_t graph = nullptr;
      hipError_t eerr = (cerr == hipSuccess) ? hipStreamEndCapture(h->gstream, &graph) : cerr;
      if (inject == 2 && eerr == hipSuccess) eerr = hipErrorUnknown;
      h->capturing = false;
      c0.stream = s_keep; c0.fstream = f_keep;
      // what failed: the launches themselves (rc: a bad argument, a HIP error of a launch -- the same call would fail eagerly too) or
      // the capture machinery (BeginCapture / EndCapture / Instantiate: the step itself is fine)
      const int enqueue_rc = rc;
      std::string why;
      if (enqueue_rc == PAM_AMD_OK && eerr != hipSuccess) why = std::string("graph capture: ") + hipGetErrorString(eerr);
      if (enqueue_rc == PAM_AMD_OK && why.empty() &&
          (inject == 3 || hipGraphInstantiate(&e.exec, graph, nullptr, nullptr, 0) != hipSuccess)) why = "hipGraphInstantiate failed";
      if (graph) (void)hipGraphDestroy(graph);
      if (enqueue_rc != PAM_AMD_OK || !why.empty()) {
        h->fct_seq = seq_keep; h->prim0 = p0_keep; h->prim1 = p1_keep;
        // the caller's stream was forked into gstream above: join it again
        (void)hipEventRecord(h->g_join, h->gstream);
        (void)hipStreamWaitEvent(h->stream, h->g_join, 0);
        if (enqueue_rc != PAM_AMD_OK) return enqueue_rc;         // (its message is in last_error; nothing ran: the launches were only recorded)
        // the capture machinery failed: drop the runtime's sticky error, run the step eagerly, say so, and stop capturing on this handle
        (void)hipGetLastError();
        h->graph_mode = 1;
        const int erc = enqueue();
        if (erc == PAM_AMD_OK) g_last_error = "warning: time_step: " + why + "; the step ran eagerly and graph replay is switched off for this handle";
        return erc;
      }
      e.prim0_after = h->prim0; e.prim1_after = h->prim1;       // (the capture has walked the buffer rotation on the host side)
      h->graphs.push_back(e);
      entry = &h->graphs.back();
    } else {
      h->fct_seq += nstages;
      h->prim0 = entry->prim0_after;
      h->prim1 = entry->prim1_after;
    }
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)h->seq_dev, seq0, 1, h->gstream));
    HIP_TRY(hipGraphLaunch(entry->exec, h->gstream));
    HIP_TRY(hipEventRecord(h->g_join, h->gstream));
    HIP_TRY(hipStreamWaitEvent(h->stream, h->g_join, 0));
    return PAM_AMD_OK;
  }
  rc = enqueue();
  // join: the caller's stream continues after every chunk (and the shared compute stream, which every chunk stream follows
  // or is followed by through the events above) has finished -- also on the error path
  if (forked) {
    int jrc = PAM_AMD_OK;
    auto join_try = [&](hipError_t e) { if (e != hipSuccess && jrc == PAM_AMD_OK) jrc = fail(PAM_AMD_ENOGPU, std::string("time_step join: ") + hipGetErrorString(e)); };
    for (auto &c : h->chunks) {
      if (c.fstream && c.fstream != c.stream) {       // flux / compute streams: order them into the chunk stream first
        join_try(hipEventRecord(c.flux_done, c.fstream));
        join_try(hipStreamWaitEvent(c.stream, c.flux_done, 0));
      }
      join_try(hipEventRecord(c.done, c.stream));
      join_try(hipStreamWaitEvent(h->stream, c.done, 0));
    }
    if (rc == PAM_AMD_OK) rc = jrc;
  }
  if (rc == PAM_AMD_OK && h->debug_mass) rc = evaluate_mass(h);
  return rc;
}

int pam_amd_awfl_init_idealized(pam_amd_awfl_t *h, const pam_amd_awfl_fields_t *fields, const char *init_data,
                                const double *vertical_midpoint_height, const double *vertical_interface_height) {
  if (!h || !init_data || !vertical_midpoint_height || !vertical_interface_height)
    return fail(PAM_AMD_EINVAL, "init_idealized: null argument");
  USE_DEVICE(h);
  const std::string kind(init_data);
  if (kind == "external") return PAM_AMD_OK;                                       // Dycore.h:1009: nothing to do
  if (kind != "thermal" && kind != "supercell") return fail(PAM_AMD_EINVAL, "ERROR: Invalid data_spec");   // Dycore.h:1002
  TracerPtrs tp;
  int rc = make_tracer_ptrs(h, fields, tp);
  if (rc) return rc;
  const Params &P = h->P;
  const EnsRange r = full_range(P);
  if (kind == "thermal") {
    ScopedTimer st(h, "init_idealized", h->stream);
    hipLaunchKernelGGL(awfl_init_thermal_kernel, cell_grid(P, r), dim3(256), 0, h->stream, P, r, h->cfg.xlen, h->cfg.ylen,
                       h->cp_d, h->p0, vertical_midpoint_height, fields->density_dry, fields->uvel, fields->vvel, fields->wvel,
                       fields->temp, tp);
    HIP_TRY(hipGetLastError());
    return PAM_AMD_OK;
  }
  // supercell: integrate the sounding column on the host (once), then fill the 3-D fields on the device
  const size_t nzn = (size_t)P.nz * P.nens;
  std::vector<double> dz(nzn), zmid(nzn), zint(nzn + P.nens);
  HIP_TRY(hipMemcpy(dz.data(), h->dz, nzn * 8, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(zmid.data(), vertical_midpoint_height, nzn * 8, hipMemcpyDefault));
  HIP_TRY(hipMemcpy(zint.data(), vertical_interface_height, (nzn + P.nens) * 8, hipMemcpyDefault));
  SupercellColumns sc = supercell_columns(dz.data(), zmid.data(), zint.data(), P.nz, P.nens, h->R_d, h->R_v, h->grav,
                                          h->gamma_d, h->C0);
  double *d_hd = nullptr, *d_hp = nullptr, *d_dv = nullptr;
  HIP_TRY(hipMalloc(&d_hd, nzn * 8));
  HIP_TRY(hipMalloc(&d_hp, nzn * 8));
  HIP_TRY(hipMalloc(&d_dv, nzn * 9 * 8));
  HIP_TRY(hipMemcpy(d_hd, sc.hy_dens.data(), nzn * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_hp, sc.hy_pres.data(), nzn * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_dv, sc.dens_vap_gll.data(), nzn * 9 * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(awfl_init_supercell_kernel, cell_grid(P, r), dim3(256), 0, h->stream, P, r, vertical_midpoint_height,
                     d_hd, d_hp, d_dv, fields->density_dry, fields->uvel, fields->vvel, fields->wvel, fields->temp, tp);
  hipError_t err = hipGetLastError();
  (void)hipStreamSynchronize(h->stream);
  (void)hipFree(d_hd); (void)hipFree(d_hp); (void)hipFree(d_dv);
  if (err != hipSuccess) return fail(PAM_AMD_ENOGPU, hipGetErrorString(err));
  return PAM_AMD_OK;
}

int pam_amd_awfl_set_kernel_timing(pam_amd_awfl_t *h, int enable) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  drop_graphs(h);
  h->timing = enable != 0;
  return PAM_AMD_OK;
}

int pam_amd_awfl_get_kernel_timing(pam_amd_awfl_t *h, const char *name, double *total_ms, long long *launches) {
  if (!h || !name || !total_ms || !launches) return fail(PAM_AMD_EINVAL, "get_kernel_timing: null argument");
  auto it = h->timers.find(name);
  if (it == h->timers.end()) { *total_ms = 0; *launches = 0; return PAM_AMD_OK; }
  drain(it->second);
  *total_ms = it->second.total_ms;
  *launches = it->second.launches;
  return PAM_AMD_OK;
}

int pam_amd_awfl_reset_kernel_timing(pam_amd_awfl_t *h) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  for (auto &kv : h->timers) drain(kv.second);
  h->timers.clear();
  return PAM_AMD_OK;
}

int pam_amd_awfl_set_flux_segment(pam_amd_awfl_t *h, int faces) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  if (faces < 1 || faces > FLUX_MAX_SPAN) return fail(PAM_AMD_EINVAL, "set_flux_segment: faces must be in [1,64]");
  drop_graphs(h);
  h->P.seg = faces;
  return PAM_AMD_OK;
}

int pam_amd_awfl_set_flux_span(pam_amd_awfl_t *h, int faces) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  if (faces < 0) return fail(PAM_AMD_EINVAL, "set_flux_span: faces must be >= 0 (0 = automatic)");
  drop_graphs(h);
  h->span_override = faces;
  return PAM_AMD_OK;
}

int pam_amd_awfl_set_ensemble_chunks(pam_amd_awfl_t *h, int chunks, int flux_lds_floor_bytes) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  if (chunks < 0 || chunks > 16) return fail(PAM_AMD_EINVAL, "set_ensemble_chunks: chunks must be in [0,16] (0 = automatic)");
  if (flux_lds_floor_bytes < 0 || flux_lds_floor_bytes > 160 * 1024)
    return fail(PAM_AMD_EINVAL, "set_ensemble_chunks: flux_lds_floor_bytes must be in [0, 163840]");
  USE_DEVICE(h);
  h->chunks_requested = chunks;
  h->flux_lds_floor = (size_t)flux_lds_floor_bytes;
  return build_chunks(h);
}

int pam_amd_awfl_set_lane_mapping(pam_amd_awfl_t *h, int yz_lanes, int x_kernels) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  if (yz_lanes < 0 || yz_lanes > 2 || x_kernels < 0 || x_kernels > 2)
    return fail(PAM_AMD_EINVAL, "set_lane_mapping: 0 = automatic, 1 = member lanes / sweep kernels, 2 = flat lanes / tile kernels");
  USE_DEVICE(h);
  const int old_l = h->lane_mode, old_x = h->xtile_mode;
  h->lane_mode = yz_lanes;
  h->xtile_mode = x_kernels;
  resolve_lane_mapping(h);
  if ((yz_lanes == 2 && !h->flat) || (x_kernels == 2 && !h->xtile)) {
    h->lane_mode = old_l; h->xtile_mode = old_x;
    resolve_lane_mapping(h);
    return fail(PAM_AMD_EINVAL, "set_lane_mapping: flat lanes / tile kernels need every field below 2^28 doubles (32-bit lane offsets)");
  }
  return build_chunks(h);
}

int pam_amd_awfl_set_x_tile(pam_amd_awfl_t *h, int row_lanes, int cells_per_tile, int lines_per_group) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  if (row_lanes < 0 || cells_per_tile < 0 || lines_per_group < 0) return fail(PAM_AMD_EINVAL, "set_x_tile: arguments must be >= 0 (0 = automatic)");
  const XTileGeom g = xtile_geometry(h->P, row_lanes, cells_per_tile, lines_per_group, h->ncu);
  if (xtile_threads(g) > 1024 || xtile_threads(g) < 1) return fail(PAM_AMD_EINVAL, "set_x_tile: a tile must fit a workgroup of 1024 lanes");
  USE_DEVICE(h);
  // the LDS bound is that of the launch (launch_xupd): it applies when the x tile kernels run AND exchange through LDS -- the shuffle
  // form stages nothing -- so the candidate geometry is resolved first and rolled back if the launch would refuse it
  const int old_w = h->xt_w, old_tc = h->xt_tc, old_lpb = h->xt_lpb;
  h->xt_w = row_lanes; h->xt_tc = cells_per_tile; h->xt_lpb = lines_per_group;
  resolve_lane_mapping(h);       // (may switch the x kernels: the ranges are rebuilt like set_lane_mapping does)
  if (h->xtile && !h->xshuf && (size_t)XT_NS * (xtile_threads(h->xg) + xtile_stage_elems(h->xg)) * sizeof(double) > 160 * 1024) {
    h->xt_w = old_w; h->xt_tc = old_tc; h->xt_lpb = old_lpb;
    resolve_lane_mapping(h);
    return fail(PAM_AMD_EINVAL, "set_x_tile: the staged tile does not fit the 160 KB of LDS");
  }
  return build_chunks(h);
}

int pam_amd_awfl_set_x_exchange(pam_amd_awfl_t *h, int mode) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  if (mode < 0 || mode > 2) return fail(PAM_AMD_EINVAL, "set_x_exchange: 0 = automatic, 1 = through LDS, 2 = wavefront shuffles");
  const int old = h->xshuf_mode;
  h->xshuf_mode = mode;
  resolve_lane_mapping(h);
  if (mode == 2 && !h->xshuf) {
    h->xshuf_mode = old;
    resolve_lane_mapping(h);
    return fail(PAM_AMD_EINVAL, "set_x_exchange: wavefront shuffles need x tile kernels whose whole periodic line lies inside one wavefront (nx * row lanes divides 64)");
  }
  return PAM_AMD_OK;
}

int pam_amd_awfl_set_flux_tile(pam_amd_awfl_t *h, int enable, int cells_per_y_tile, int levels_per_z_tile) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  if (cells_per_y_tile < 0 || levels_per_z_tile < 0) return fail(PAM_AMD_EINVAL, "set_flux_tile: tile sizes must be >= 0 (0 = automatic)");
  if (enable < 0 || enable > 2) return fail(PAM_AMD_EINVAL, "set_flux_tile: 0 = automatic, 1 = flat-lane sweeps, 2 = tile kernel");
  h->ftile_mode = enable;
  h->ft_tc_y = cells_per_y_tile;
  h->ft_tc_z = levels_per_z_tile;
  resolve_lane_mapping(h);
  return PAM_AMD_OK;
}

int pam_amd_awfl_set_flux_tile_parts(pam_amd_awfl_t *h, int mode) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  if (mode < 0 || mode > 2) return fail(PAM_AMD_EINVAL, "set_flux_tile_parts: 0 = automatic, 1 = behind each other (one workgroup per tile), 2 = beside each other");
  h->ftile_parts_mode = mode;
  resolve_lane_mapping(h);
  return PAM_AMD_OK;
}

int pam_amd_awfl_set_tile_state_parts(pam_amd_awfl_t *h, int mode) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  if (mode < 0 || mode > 2) return fail(PAM_AMD_EINVAL, "set_tile_state_parts: 0 = automatic, 1 = one lane finishes a cell's whole state, 2 = three parts beside each other");
  h->tile_state_parts_mode = mode;
  resolve_lane_mapping(h);
  return PAM_AMD_OK;
}

int pam_amd_awfl_set_tile_fusion(pam_amd_awfl_t *h, int mode) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  if (mode < 0 || mode > 3)
    return fail(PAM_AMD_EINVAL, "set_tile_fusion: 0 = automatic, 1 = separate launches, 2 = inside the x tile kernel (tracer phase 1 behind the state pass), 3 = inside, tracer phase 1 in workgroups beside it");
  h->tile_pressure_mode = mode;
  resolve_lane_mapping(h);
  return PAM_AMD_OK;
}

int pam_amd_awfl_set_graph_replay(pam_amd_awfl_t *h, int mode) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  if (mode < 0 || mode > 2) return fail(PAM_AMD_EINVAL, "set_graph_replay: 0 = automatic, 1 = off, 2 = on");
  USE_DEVICE(h);
  if (h->gstream) (void)hipStreamSynchronize(h->gstream);
  drop_graphs(h);
  h->graph_mode = mode;
  if (mode == 2 && !h->gstream) {
    HIP_TRY(hipStreamCreateWithFlags(&h->gstream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&h->g_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->g_join, hipEventDisableTiming));
    HIP_TRY(hipMalloc(&h->seq_dev, sizeof(int)));
  }
  return PAM_AMD_OK;
}

int pam_amd_awfl_debug_fail_next_capture(pam_amd_awfl_t *h, int which) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  if (which < 0 || which > 3) return fail(PAM_AMD_EINVAL, "debug_fail_next_capture: 0 none, 1 BeginCapture, 2 EndCapture, 3 Instantiate");
  h->fail_next_capture = which;
  return PAM_AMD_OK;
}

int pam_amd_awfl_set_launch_tuning(long long want_units, long long two_phase_below, long long split_below) {
  if (want_units > 0) g_want_units = want_units;
  if (two_phase_below >= 0) g_two_phase_below = two_phase_below;
  if (split_below >= 0) g_split_below = split_below;
  return PAM_AMD_OK;
}

int pam_amd_awfl_set_handle_launch_tuning(pam_amd_awfl_t *h, long long want_units, long long two_phase_below, long long split_below) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  USE_DEVICE(h);
  if (want_units > 0) h->want_units = want_units;
  if (two_phase_below >= 0) h->two_phase_below = two_phase_below;
  if (split_below >= 0) h->split_below = split_below;
  return build_chunks(h);      // (the automatic range count looks at the spans; drains the handle's streams and drops captured graphs)
}

int pam_amd_awfl_set_tracer_grouping(pam_amd_awfl_t *h, int tracers_per_wavefront, int prefetch) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  if (tracers_per_wavefront != 0 && tracers_per_wavefront != 1 && tracers_per_wavefront != 2 && tracers_per_wavefront != 4)
    return fail(PAM_AMD_EINVAL, "set_tracer_grouping: 0 (automatic), 1, 2 or 4 tracers per wavefront");
  if (prefetch && tracers_per_wavefront != 2) return fail(PAM_AMD_EINVAL, "set_tracer_grouping: the one-trip-ahead form exists for pairs only");
  drop_graphs(h);
  h->tracers_per_wave = tracers_per_wavefront;
  h->tracer_prefetch = prefetch != 0;
  return PAM_AMD_OK;
}

int pam_amd_awfl_get_lane_mapping(const pam_amd_awfl_t *h, int *yz_flat, int *x_tiles, int *flat_cells, int geom[6]) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  if (yz_flat) *yz_flat = h->flat ? (h->ftile ? 2 : 1) : 0;
  if (x_tiles) *x_tiles = h->xtile ? 1 : 0;
  if (flat_cells) *flat_cells = h->P.flat_cells;
  if (x_tiles && h->xtile && h->xshuf) *x_tiles = 2;
  if (geom) { geom[0] = h->xg.W; geom[1] = h->xg.nmb; geom[2] = h->xg.tc; geom[3] = h->xg.halo; geom[4] = h->xg.ntl; geom[5] = h->xg.lpb; }
  return PAM_AMD_OK;
}

int pam_amd_awfl_set_debug_conservation(pam_amd_awfl_t *h, int enable) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  USE_DEVICE(h);
  drop_graphs(h);
  if (enable && !h->mass_dev) HIP_TRY(hipMalloc(&h->mass_dev, (size_t)2 * (h->P.nt + 2) * h->P.nens * sizeof(double)));
  h->debug_mass = enable != 0;
  h->mass_violations = 0; h->mass_max_rel = 0.0; h->mass_worst_var = h->mass_worst_member = -1;
  h->mass_report.clear();
  return PAM_AMD_OK;
}

int pam_amd_awfl_get_conservation(pam_amd_awfl_t *h, int *violations, double *max_rel_diff, int *worst_variable, int *worst_member) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  if (!h->debug_mass) return fail(PAM_AMD_ESTATE, "get_conservation: the check is off (pam_amd_awfl_set_debug_conservation)");
  if (violations) *violations = h->mass_violations;
  if (max_rel_diff) *max_rel_diff = h->mass_max_rel;
  if (worst_variable) *worst_variable = h->mass_worst_var;
  if (worst_member) *worst_member = h->mass_worst_member;
  return PAM_AMD_OK;
}

const char *pam_amd_awfl_conservation_report(const pam_amd_awfl_t *h) { return h ? h->mass_report.c_str() : ""; }

int pam_amd_awfl_debug_inject_mass_fault(pam_amd_awfl_t *h, int variable, int k, int j, int i, int member, double factor) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  const Params &P = h->P;
  if (variable < 0 || variable > P.nt + 1 || k < 0 || k >= P.nz || j < 0 || j >= P.ny || i < 0 || i >= P.nx || member < 0 || member >= P.nens)
    return fail(PAM_AMD_EINVAL, "debug_inject_mass_fault: index out of range");
  h->fault.armed = true; h->fault.ivar = variable; h->fault.k = k; h->fault.j = j; h->fault.i = i; h->fault.e = member; h->fault.factor = factor;
  return PAM_AMD_OK;
}

int pam_amd_awfl_set_tail_fusion(pam_amd_awfl_t *h, int mode) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  if (mode < 0 || mode > 2) return fail(PAM_AMD_EINVAL, "set_tail_fusion: 0 = automatic, 1 = three launches, 2 = one launch");
  drop_graphs(h);
  h->tail_fuse_mode = mode;
  return PAM_AMD_OK;
}

int pam_amd_awfl_set_yz_fold(pam_amd_awfl_t *h, int mode) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  if (mode < 0 || mode > 2) return fail(PAM_AMD_EINVAL, "set_yz_fold: 0 = automatic, 1 = off (the x-sweep loads the y and the z differences), 2 = on");
  USE_DEVICE(h);
  const int old = h->fold_mode;
  h->fold_mode = mode;
  resolve_lane_mapping(h);
  if (mode == 2 && !h->P.yz_fold) {
    h->fold_mode = old;
    resolve_lane_mapping(h);
    return fail(PAM_AMD_EINVAL, "set_yz_fold: the fold exists for 3-D grids swept with member lanes (y, z and x sweep kernels)");
  }
  return PAM_AMD_OK;
}

int pam_amd_awfl_set_range_schedule(pam_amd_awfl_t *h, int independent) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  h->independent_ranges = independent != 0;
  USE_DEVICE(h);
  return build_chunks(h);
}

int pam_amd_awfl_set_fused_stage(pam_amd_awfl_t *h, int enable) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  if (enable && !h->fused_supported) return fail(PAM_AMD_EINVAL, "set_fused_stage: not available on this handle");
  const bool changed = h->fused != (enable != 0);
  h->fused = enable != 0;
  if (changed) {   // the automatic range count and the streams a range needs depend on the stage structure
    USE_DEVICE(h);
    return build_chunks(h);
  }
  return PAM_AMD_OK;
}

int pam_amd_awfl_debug_get_buffer(pam_amd_awfl_t *h, const char *name, double **device_ptr, size_t *nelem) {
  if (!h || !name || !device_ptr || !nelem) return fail(PAM_AMD_EINVAL, "debug_get_buffer: null argument");
  const std::string k(name);
  if (k == "prim0") { *device_ptr = h->prim0; *nelem = h->n_prim; }
  else if (k == "prim1") { *device_ptr = h->prim1; *nelem = h->n_prim; }
  else if (k == "prim2") { *device_ptr = h->prim2; *nelem = h->prim2 ? h->n_prim : 0; }
  else if (k == "flux_x") { *device_ptr = h->flux_x; *nelem = h->n_flux_xy; }
  else if (k == "flux_y") { *device_ptr = h->flux_y; *nelem = h->P.sim2d ? 0 : h->n_flux_xy; }
  else if (k == "flux_z") { *device_ptr = h->flux_z; *nelem = h->n_flux_z; }
  else if (k == "seed") { *device_ptr = h->seed; *nelem = h->n_seed; }
  else if (k == "mult") { *device_ptr = h->mult; *nelem = h->n_seed; }
  else return fail(PAM_AMD_EINVAL, "debug_get_buffer: unknown buffer " + k);
  return PAM_AMD_OK;
}

__global__ void __launch_bounds__(256) awfl_pow_kat_kernel(Params P, const double *__restrict__ x, int n, double y, double *__restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) out[t] = pow_pos(P, x[t], y);
}

int pam_amd_awfl_debug_pow(pam_amd_awfl_t *h, const double *x, int n, double y, double *out) {
  if (!h || !x || !out || n < 0) return fail(PAM_AMD_EINVAL, "debug_pow: bad argument");
  USE_DEVICE(h);
  if (n == 0) return PAM_AMD_OK;
  hipLaunchKernelGGL(awfl_pow_kat_kernel, dim3(nblocks(n, 256)), dim3(256), 0, h->stream, h->P, x, n, y, out);
  HIP_TRY(hipGetLastError());
  return PAM_AMD_OK;
}

int pam_amd_awfl_debug_fct_rows(pam_amd_awfl_t *h, long long *rows_flagged, long long *rows_total, int *any_flagged) {
  if (!h || !rows_flagged || !rows_total || !any_flagged) return fail(PAM_AMD_EINVAL, "debug_fct_rows: null argument");
  USE_DEVICE(h);
  std::vector<int> flags(h->n_fct_flags + h->n_fct_any);
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(flags.data(), h->fct_flags, flags.size() * sizeof(int), hipMemcpyDeviceToHost));
  long long n = 0;
  for (size_t i = 0; i < h->n_fct_flags; i++) n += (flags[i] == h->fct_seq) ? 1 : 0;
  *rows_flagged = n;
  *rows_total = (long long)h->n_fct_flags;
  *any_flagged = 0;
  for (size_t b = 0; b < h->n_fct_any; b++) *any_flagged |= (flags[h->n_fct_flags + b] == h->fct_seq) ? 1 : 0;
  return PAM_AMD_OK;
}

int pam_amd_awfl_debug_weno(pam_amd_awfl_t *h, int level, const double *stencils, int n, double *left, double *right) {
  if (!h || !stencils || !left || !right || n < 0) return fail(PAM_AMD_EINVAL, "debug_weno: bad argument");
  if (level > h->P.nz + 1) return fail(PAM_AMD_EINVAL, "debug_weno: level must be < nz+2 (negative = uniform-grid constants)");
  USE_DEVICE(h);
  if (n == 0) return PAM_AMD_OK;
  hipLaunchKernelGGL(awfl_weno_kat_kernel, dim3(nblocks(n, 256)), dim3(256), 0, h->stream, h->P, level, stencils, n, left, right);
  HIP_TRY(hipGetLastError());
  return PAM_AMD_OK;
}

// One tendency stage on its own: stage 1 of a sub-step of length dt_dyn from the resident state (a forward-Euler step
// prim0 -> new prim0; the FCT seed is left as stage 1 leaves it), with whichever stage structure is selected.
int pam_amd_awfl_debug_stage(pam_amd_awfl_t *h, double dt_dyn) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  if (!h->hydro_declared) return fail(PAM_AMD_ESTATE, "debug_stage: declare_current_profile_as_hydrostatic first");
  USE_DEVICE(h);
  const EnsRange r = full_range(h->P);
  int rc;
  if ((rc = next_fct_stage(h))) return rc;
  if ((rc = launch_flux(h, h->prim0, r, h->stream, h->fused ? 6 : 7, h->fused))) return rc;
  bool fuse_tail = false;
  if (h->fused && (rc = launch_xupd<1>(h, h->prim0, h->prim0, h->prim1, dt_dyn, dt_dyn, r, h->stream, tail_fusable(h), &fuse_tail))) return rc;
  if (!h->fused && (rc = launch_fct(h, dt_dyn, r, h->stream))) return rc;
  if (h->fused) rc = fuse_tail ? PAM_AMD_OK : launch_tail<1>(h, h->prim0, h->prim0, h->prim1, dt_dyn, r, h->stream);
  else rc = launch_update<1>(h, h->prim0, h->prim0, h->prim1, dt_dyn, r, h->stream);
  if (rc) return rc;
  std::swap(h->prim0, h->prim1);
  return PAM_AMD_OK;
}

int pam_amd_awfl_debug_flux_stage(pam_amd_awfl_t *h, double dt) {
  if (!h) return fail(PAM_AMD_EINVAL, "null handle");
  USE_DEVICE(h);
  int rc;
  if ((rc = launch_flux(h, h->prim0, full_range(h->P), h->stream))) return rc;
  if ((rc = next_fct_stage(h))) return rc;
  return launch_fct(h, dt, full_range(h->P), h->stream);
}

}  // extern "C"
