// awfl_emu.cpp -- HOST EMULATION of the HIP kernel bodies.  TEST INFRASTRUCTURE ONLY (never shipped, never
// linked into libpam_amd_awfl.so, never used by bench.py or the product package).
//
// It compiles pam_amd/csrc/awfl_device.h with g++ and runs every kernel body once per (block, thread) in plain
// loops, with the launch geometry of pam_amd/csrc/awfl_kernels.hip, so that the index logic (segments, periodic
// wrap, ghosts, FCT seam, RK aliasing) can be compared with the oracle on a machine without a GPU.
// Differences to the device build: true division instead of v_rcp_f64+Newton, glibc pow, and whatever
// contraction g++ applies -- i.e. rounding-level only.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../pam_amd/csrc/awfl_device.h"
#include "../../pam_amd/csrc/awfl_vertical.h"

using namespace pama;

struct Emu {
  Params P;
  std::vector<double> prim0, prim1, prim2, fx, fy, fz, seed, mult, dz, grav_var, hy_dens, hy_pres, vz, rdz;
  VerticalTables vt;
  PowTab pow_tab;
  int span = 0;   // faces per thread in the flux sweep (0 = whole line)
  int xtr_split = 0;   // fused stage: tracers 1.. swept in a launch of their own (awfl_xtr_kernel) instead of inline
  int fused = 0;  // stage structure of awfl_kernels.hip: 1 = flux(y,z) -> fused x-sweep + state update -> FCT -> tracer update
  int flat = 0;   // fused stage: the y/z sweeps run with flat (x, member) lanes (flat_lane) and the fix-up with a lane per cell
  int xtile = 0;  // fused stage: the x direction runs as tile kernels (a lane per cell; xtile_* bodies) instead of sweeps
  int xt_w = 0, xt_tc = 0, xt_lpb = 0;   // tile geometry overrides
  int tile_pressure = 0;   // x tile kernels: the pressure pass inside xtile_state_finish (no pressure_tail_body pass)
  int ftile = 0, ft_tc_y = 0, ft_tc_z = 0;   // flat lanes: the y/z fluxes as tile kernels (ftile_* bodies); cells / levels per tile
  std::vector<int> fct_flags;   // row flags of the FCT multiplier (FctRows)
  int fct_seq = 0;
};

static FctRows fct_rows(Emu *h) {   // layout as on the device: rows, the "any" word, line flags
  FctRows r;
  const size_t nrows = (size_t)h->P.nt * fct_rows_per_tracer(h->P);
  const size_t nany = (size_t)((h->P.nens + 63) >> 6);
  r.flags = h->fct_flags.data(); r.any = h->fct_flags.data() + nrows; r.lines = h->fct_flags.data() + nrows + nany;
  r.seq = h->fct_seq; r.sparse_store = 0;
  return r;
}

// awfl_flux_tile_kernel: the phases of a workgroup one after the other over all its lanes, LDS exchange in between
template <int DIR, bool VZ>
static void flux_tile_dir(Emu *h, const double *prim, double *flux, int tc_req) {
  const Params &P = h->P;
  const FTileGeom G = ftile_geometry(P, DIR, tc_req);
  const int rows = ftile_rows(G), T = ftile_threads(G);
  int grp[64][FT_NG];
  const int ngroups = ftile_groups(P, DIR, grp, 64);
  const int gx = G.nch * G.ntl, gy = (DIR == 1) ? (P.nz + G.lpb - 1) / G.lpb : 1;
  const int ncomp_a = (DIR == 1) ? 1 : 2;
  struct Lane { FLane X; double L[FT_NG], R[FT_NG], F[FT_NG], ruf, fn; };
  std::vector<Lane> st(T);
  std::vector<double> ldsR[2] = {std::vector<double>((size_t)FT_NG * T), std::vector<double>((size_t)FT_NG * T)}, ldsF((size_t)4 * T);
  for (int by = 0; by < gy; by++)
    for (int bx = 0; bx < gx; bx++) {
      for (auto &v : ldsR) std::fill(v.begin(), v.end(), NAN);
      std::fill(ldsF.begin(), ldsF.end(), NAN);
      for (int tz = 0; tz < G.lpb; tz++)
        for (int ty = 0; ty < rows; ty++)
          for (int tx = 0; tx < G.W; tx++) {
            const int t = (tz * rows + ty) * G.W + tx;
            Lane &l = st[t];
            l.X = ftile_lane<DIR>(P, G, bx, by, tx, ty, tz);
            if (l.X.slot != t) abort();
            l.ruf = l.fn = 0.0;
            if (!l.X.poly) continue;
            ftile_acoustic_polys<DIR, VZ>(P, prim, l.X, l.L, l.R);
            for (int n = 0; n < FT_NG; n++) ldsR[0][(size_t)n * T + t] = l.R[n];
          }
      for (int t = 0; t < T; t++)
        if (st[t].X.face) {
          Lane &l = st[t];
          for (int n = 0; n < FT_NG; n++) l.R[n] = ldsR[0][(size_t)n * T + l.X.slot_l];
          ftile_acoustic_face<DIR>(P, flux, l.X, l.L, l.R, l.ruf, l.fn);
          ldsF[t] = l.fn;
        }
      for (int g = 0; g < ngroups; g++) {
        std::vector<double> &buf = ldsR[(g + 1) & 1];
        int nf = 0;
        for (int n = 0; n < FT_NG; n++) nf += grp[g][n] >= 0 ? 1 : 0;
        for (int t = 0; t < T; t++)
          if (st[t].X.poly) {
            ftile_adv_polys<DIR, VZ>(P, prim, st[t].X, grp[g], nf, st[t].L, st[t].R);
            for (int n = 0; n < FT_NG; n++) buf[(size_t)n * T + t] = st[t].R[n];
          }
        for (int t = 0; t < T; t++)
          if (st[t].X.face) {
            Lane &l = st[t];
            for (int n = 0; n < FT_NG; n++) l.R[n] = buf[(size_t)n * T + l.X.slot_l];
            ftile_adv_face<DIR>(P, flux, l.X, grp[g], nf, l.L, l.R, l.ruf, l.F);
          }
        if (g == 0) {
          for (int t = 0; t < T; t++)
            if (st[t].X.face)
              for (int n = 0; n < FT_NG; n++) ldsF[(size_t)(1 + n) * T + t] = st[t].F[n];
          for (int t = 0; t < T; t++)
            if (st[t].X.pay) {
              Lane &l = st[t];
              ftile_store_diff<DIR>(P, flux, l.X, ncomp_a, l.fn, ldsF[l.X.slot_r]);
              for (int n = 0; n < FT_NG; n++)
                if (grp[g][n] >= 0 && grp[g][n] < 4) ftile_store_diff<DIR>(P, flux, l.X, grp[g][n], l.F[n], ldsF[(size_t)(1 + n) * T + l.X.slot_r]);
            }
        }
      }
    }
}

static void flux_launch(Emu *h, const double *prim, int sweeps = 7, bool diff = false) {
  const Params &P = h->P;
  for (int dir = 0; dir < 3; dir++) {
    if (dir == 1 && P.sim2d) continue;
    if (!((sweeps >> dir) & 1)) continue;
    const int nfaces = dir == 0 ? P.nx : (dir == 1 ? P.ny : P.nz + 1);
    const int nlines = dir == 0 ? P.nz * P.ny : (dir == 1 ? P.nz * P.nx : P.ny * P.nx);
    // span as in awfl_kernels.hip::choose_span with an override (h->span; 0 = whole line)
    const int pieces = (nfaces + FLUX_MAX_SPAN - 1) / FLUX_MAX_SPAN;
    int span = h->span > 0 ? (h->span < FLUX_MAX_SPAN ? h->span : FLUX_MAX_SPAN) : (nfaces + pieces - 1) / pieces;
    if (diff && dir == 1) span = nfaces;                 // difference form: periodic lines are swept whole
    const int nspan = (nfaces + span - 1) / span;
    double *fl = dir == 0 ? h->fx.data() : (dir == 1 ? h->fy.data() : h->fz.data());
    if (h->flat && h->ftile && diff && dir != 0) {     // awfl_flux_tile_kernel: a lane per cell
      if (dir == 1) { if (P.vz_per_ens) flux_tile_dir<1, true>(h, prim, fl, h->ft_tc_y); else flux_tile_dir<1, false>(h, prim, fl, h->ft_tc_y); }
      else { if (P.vz_per_ens) flux_tile_dir<2, true>(h, prim, fl, h->ft_tc_z); else flux_tile_dir<2, false>(h, prim, fl, h->ft_tc_z); }
      continue;
    }
    if (h->flat && diff && dir != 0) {     // awfl_flux_kernel<., true, FLAT>: a lane per item of the sweep's flat index space
      const long long items = flat_items(P, dir);
      for (int sp = 0; sp < nspan; sp++)
        for (long long q = 0; q < items; q++) {
          const int f0 = sp * span;
          if (dir == 1) {
            if (P.vz_per_ens) flux_line_body<1, true, true>(P, prim, fl, flat_lane<1>(P, (unsigned)q), f0, span);
            else flux_line_body<1, false, true>(P, prim, fl, flat_lane<1>(P, (unsigned)q), f0, span);
          } else {
            if (P.vz_per_ens) flux_line_body<2, true, true>(P, prim, fl, flat_lane<2>(P, (unsigned)q), f0, span);
            else flux_line_body<2, false, true>(P, prim, fl, flat_lane<2>(P, (unsigned)q), f0, span);
          }
        }
      continue;
    }
    // one wavefront per (line, block of 64 members, span); lanes = members
    for (int line = 0; line < nlines; line++)
      for (int sp = 0; sp < nspan; sp++)
        for (int e = 0; e < P.nens; e++) {
          const int f0 = sp * span;
          if (dir == 0) {
            if (P.vz_per_ens) flux_line_body<0, true, false>(P, prim, fl, line, e, f0, span);
            else flux_line_body<0, false, false>(P, prim, fl, line, e, f0, span);
          } else if (dir == 1) {
            if (P.vz_per_ens) { if (diff) flux_line_body<1, true, true>(P, prim, fl, line, e, f0, span); else flux_line_body<1, true, false>(P, prim, fl, line, e, f0, span); }
            else { if (diff) flux_line_body<1, false, true>(P, prim, fl, line, e, f0, span); else flux_line_body<1, false, false>(P, prim, fl, line, e, f0, span); }
          } else if (diff && P.yz_fold) {      // the z sweep of a folded stage (the y sweep has run: dir 1 comes first)
            if (P.vz_per_ens) flux_line_body<2, true, true, true>(P, prim, fl, line, e, f0, span, -1, h->fy.data());
            else flux_line_body<2, false, true, true>(P, prim, fl, line, e, f0, span, -1, h->fy.data());
          } else {
            if (P.vz_per_ens) { if (diff) flux_line_body<2, true, true>(P, prim, fl, line, e, f0, span); else flux_line_body<2, true, false>(P, prim, fl, line, e, f0, span); }
            else { if (diff) flux_line_body<2, false, true>(P, prim, fl, line, e, f0, span); else flux_line_body<2, false, false>(P, prim, fl, line, e, f0, span); }
          }
        }
  }
}

// the stage's flag value is drawn by the caller BEFORE the stage (h->fct_seq++), as next_fct_stage() does on the device
static void fct_launch(Emu *h, double dt) {
  const Params &P = h->P;
  for (long long idx = 0; idx < P.ncell; idx++)
    fct_mult_body(P, h->fx.data(), h->fy.data(), h->fz.data(), h->seed.data(), h->mult.data(), fct_rows(h), dt, cell_of(P, idx), 0);
}
// What the device does in the fused stage: water vapour's multipliers of a row no member of which was limited are not stored by the
// state pass.  Emulated by poisoning them after the sweeps: a fix-up that loads one of them anyway produces NaN.
static void poison_unflagged_mult(Emu *h) {
  const Params &P = h->P;
  const long long nrows = fct_rows_per_tracer(P);
  for (int t = P.idWV; t <= P.idWV; t++)     // water vapour only: the further tracers' multipliers are a complete field (own_multiplier_cell<true>)
    for (long long idx = 0; idx < P.ncell; idx++) {
      const CellId c = cell_of(P, idx);
      if (h->fct_flags[(size_t)t * nrows + fct_row(P, c.k, c.j, c.i, c.e)] != h->fct_seq) h->mult[(size_t)t * P.ncell + idx] = NAN;
    }
}

template <int STAGE>
static void update_launch(Emu *h, const double *in, const double *p0, double *out, double dt) {
  for (long long idx = 0; idx < h->P.ncell; idx++)
    update_body<STAGE>(h->P, in, p0, out, h->fx.data(), h->fy.data(), h->fz.data(), h->mult.data(), fct_rows(h), h->seed.data(), dt,
                       cell_of(h->P, idx));
}

// awfl_xupd_tile_kernel / awfl_xtr_tile_kernel: the workgroup's phases run one after the other over all its lanes, with the LDS
// exchange between them (what __syncthreads orders on the device)
template <int STAGE>
static void xupd_tile_launch(Emu *h, const double *in, const double *p0, double *out, double dt, double dt_stage) {
  const Params &P = h->P;
  const XTileGeom G = xtile_geometry(P, h->xt_w, h->xt_tc, h->xt_lpb);
  const int rows = xtile_rows(G), T = xtile_threads(G);
  const int gx = G.ntl * G.nmb, gy = (P.nz * P.ny + G.lpb - 1) / G.lpb;
  const int TS = xtile_stage_elems(G);
  struct Lane { XLane X; double L[XT_NS], R[XT_NS], cen[6], F[XT_NF], own[XT_NS]; };
  std::vector<Lane> st(T);
  std::vector<double> stage((size_t)XT_NS * TS), lds((size_t)XT_NS * T);
  const FctRows fr = fct_rows(h);
  const int npairs = (P.nt - 1 + 1) / 2;
  int sfields[XT_NS];
  xtile_state_fields(P, sfields);
  for (int by = 0; by < gy; by++)
    for (int bx = 0; bx < gx; bx++) {
      std::fill(lds.begin(), lds.end(), NAN);
      std::fill(stage.begin(), stage.end(), NAN);
      for (int tz = 0; tz < G.lpb; tz++)
        for (int ty = 0; ty < rows; ty++)
          for (int tx = 0; tx < G.W; tx++) {
            const int t = (tz * rows + ty) * G.W + tx;
            Lane &l = st[t];
            l.X = xtile_lane(P, G, bx, by, tx, ty, tz);
            if (l.X.slot != t) abort();
            xtile_stage<XT_NS>(P, in, l.X, sfields, stage.data(), TS, l.own);
          }
      for (int t = 0; t < T; t++)
        if (st[t].X.poly) {
          Lane &l = st[t];
          xtile_state_polys(P, l.X, stage.data(), TS, l.own, l.L, l.R, l.cen);
          for (int f = 0; f < XT_NS; f++) lds[(size_t)f * T + l.X.slot] = l.R[f];
        }
      for (int t = 0; t < T; t++)
        if (st[t].X.face)
          for (int f = 0; f < XT_NS; f++) st[t].R[f] = lds[(size_t)f * T + st[t].X.slot_l];
      std::fill(stage.begin(), stage.end(), NAN);      // (the staged tile is dead: the face fluxes take its place)
      for (int t = 0; t < T; t++)
        if (st[t].X.face) {
          xtile_state_face(P, h->fx.data(), st[t].X, st[t].L, st[t].R, st[t].X.upd, st[t].F);
          for (int f = 0; f < XT_NF; f++) stage[(size_t)f * T + st[t].X.slot] = st[t].F[f];
        }
      for (int t = 0; t < T; t++)
        if (st[t].X.upd) {
          double Fhi[XT_NF];
          for (int f = 0; f < XT_NF; f++) Fhi[f] = stage[(size_t)f * T + st[t].X.slot_r];
          xtile_state_finish<STAGE>(P, in, p0, out, h->fy.data(), h->fz.data(), h->seed.data(), h->mult.data(), fr, st[t].X, st[t].F,
                                    Fhi, st[t].cen, dt, dt_stage, h->tile_pressure != 0);
        }
    }
  auto tracer_phase = [&](auto phase_tag) {
    constexpr int PHASE = decltype(phase_tag)::value;
    struct TL { XLane X; double L[2], R[2], cen[2], F[2], own[2]; };
    std::vector<TL> tl(T);
    auto run_pair = [&](auto nf_tag, const int *fa) {
      constexpr int NF = decltype(nf_tag)::value;
      int fields[NF];
      for (int n = 0; n < NF; n++) fields[n] = P_U + fa[n];
      for (int by = 0; by < gy; by++)
        for (int bx = 0; bx < gx; bx++) {
          std::fill(lds.begin(), lds.end(), NAN);
          std::fill(stage.begin(), stage.end(), NAN);
          for (int tz = 0; tz < G.lpb; tz++)
            for (int ty = 0; ty < rows; ty++)
              for (int tx = 0; tx < G.W; tx++) {
                TL &l = tl[(tz * rows + ty) * G.W + tx];
                l.X = xtile_lane(P, G, bx, by, tx, ty, tz);
                xtile_stage<NF>(P, in, l.X, fields, stage.data(), TS, reinterpret_cast<double (&)[NF]>(l.own));
              }
          for (int t = 0; t < T; t++)
            if (tl[t].X.poly) {
              TL &l = tl[t];
              xtile_tracer_polys<NF>(P, l.X, stage.data(), TS, reinterpret_cast<double (&)[NF]>(l.own), reinterpret_cast<double (&)[NF]>(l.L),
                                     reinterpret_cast<double (&)[NF]>(l.R), reinterpret_cast<double (&)[NF]>(l.cen));
              for (int f = 0; f < NF; f++) lds[(size_t)f * T + l.X.slot] = l.R[f];
            }
          for (int t = 0; t < T; t++)
            if (tl[t].X.face)
              for (int f = 0; f < NF; f++) tl[t].R[f] = lds[(size_t)f * T + tl[t].X.slot_l];
          std::fill(stage.begin(), stage.end(), NAN);
          for (int t = 0; t < T; t++)
            if (tl[t].X.face) {
              TL &l = tl[t];
              xtile_tracer_face<NF>(P, h->fx.data(), l.X, reinterpret_cast<double (&)[NF]>(l.L), reinterpret_cast<double (&)[NF]>(l.R),
                                    reinterpret_cast<double (&)[NF]>(l.F));
              for (int f = 0; f < NF; f++) stage[(size_t)f * T + l.X.slot] = l.F[f];
            }
          for (int t = 0; t < T; t++)
            if (tl[t].X.upd) {
              TL &l = tl[t];
              double Fhi[NF];
              for (int f = 0; f < NF; f++) Fhi[f] = stage[(size_t)f * T + l.X.slot_r];
              xtile_tracer_finish<NF, STAGE, PHASE>(P, in, p0, out, h->fy.data(), h->fz.data(), h->seed.data(), h->mult.data(), fr, l.X, fa,
                                                    reinterpret_cast<double (&)[NF]>(l.F), Fhi, reinterpret_cast<double (&)[NF]>(l.cen), dt, dt_stage);
            }
        }
    };
    for (int pair = 0; pair < npairs; pair++) {
      const int fa[2] = {4 + further_tracer(P, 2 * pair), 4 + further_tracer(P, 2 * pair + 1)};
      if (2 * pair + 1 < P.nt - 1) run_pair(std::integral_constant<int, 2>{}, fa);
      else run_pair(std::integral_constant<int, 1>{}, fa);
    }
  };
  if (npairs > 0) tracer_phase(std::integral_constant<int, 1>{});
  poison_unflagged_mult(h);
  if (npairs > 0) tracer_phase(std::integral_constant<int, 2>{});
  if (npairs > 0 && h->tile_pressure) {      // the fix-up slice of the phase-2 tile launch (small ensembles)
    bool any = false;
    for (int b = 0; b < ((P.nens + 63) >> 6); b++) any = any || fr.any[b] == fr.seq;
    if (any)
      for (long long idx = 0; idx < P.ncell; idx++)
        tracer_fixup_cell_body<STAGE>(P, in, p0, out, h->fx.data(), h->fy.data(), h->fz.data(), h->mult.data(), fr, h->seed.data(), dt, P.idWV, cell_of(P, idx));
  }
}

// launch geometry of awfl_xupd_kernel: one wavefront per (x line, block of 64 members); lanes = members
template <int STAGE>
static void xupd_launch(Emu *h, const double *in, const double *p0, double *out, double dt, double dt_stage) {
  const Params &P = h->P;
  if (h->xtile) { xupd_tile_launch<STAGE>(h, in, p0, out, dt, dt_stage); return; }
  const int span = h->span > 0 ? h->span : P.nx, nspan = (P.nx + span - 1) / span;   // emu_set_span cuts the x lines too
  for (int line = 0; line < P.nz * P.ny; line++)
    for (int sp = 0; sp < nspan; sp++)
      for (int e = 0; e < P.nens; e++) {
        if (P.yz_fold)
          flux_x_update_body<STAGE, true>(P, in, p0, out, h->fx.data(), h->fy.data(), h->fz.data(), h->seed.data(), h->mult.data(),
                                          fct_rows(h), line, e, sp * span, span, dt, dt_stage, h->xtr_split == 0);
        else
          flux_x_update_body<STAGE, false>(P, in, p0, out, h->fx.data(), h->fy.data(), h->fz.data(), h->seed.data(), h->mult.data(),
                                           fct_rows(h), line, e, sp * span, span, dt, dt_stage, h->xtr_split == 0);
      }
  // awfl_xtr_kernel: phase 1 of the further tracers in a launch of its own (one wavefront per (span, pair)) unless it ran inline;
  // then -- always a launch of its own -- phase 2 (the multipliers of unflagged rows are poisoned in between: the device does not
  // store them)
  const int npairs = (P.nt - 1 + 1) / 2;
  for (int phase = h->xtr_split ? 1 : 2; phase <= 2; phase++) {
    if (phase == 2) poison_unflagged_mult(h);
    for (int line = 0; line < P.nz * P.ny; line++)
      for (int sp = 0; sp < nspan; sp++)
        for (int pair = 0; pair < npairs; pair++)
          for (int e = 0; e < P.nens; e++) {
            const int fa[2] = {4 + further_tracer(P, 2 * pair), 4 + further_tracer(P, 2 * pair + 1)};
            double *fxp = h->fx.data(), *fyp = h->fy.data(), *fzp = h->fz.data(), *sdp = h->seed.data(), *mtp = h->mult.data();
            if (2 * pair + 1 < P.nt - 1) {
              if (phase == 1) x_tracer_sweep<2, STAGE, 1>(P, in, p0, out, fxp, fyp, fzp, sdp, mtp, fct_rows(h), line, e, sp * span, span, fa, dt, dt_stage, false, 0.0);
              else x_tracer_sweep<2, STAGE, 2>(P, in, p0, out, fxp, fyp, fzp, sdp, mtp, fct_rows(h), line, e, sp * span, span, fa, dt, dt_stage, false, 0.0);
            } else {
              if (phase == 1) x_tracer_sweep<1, STAGE, 1>(P, in, p0, out, fxp, fyp, fzp, sdp, mtp, fct_rows(h), line, e, sp * span, span, fa, dt, dt_stage, false, 0.0);
              else x_tracer_sweep<1, STAGE, 2>(P, in, p0, out, fxp, fyp, fzp, sdp, mtp, fct_rows(h), line, e, sp * span, span, fa, dt, dt_stage, false, 0.0);
            }
          }
  }
  if (npairs == 0) poison_unflagged_mult(h);
}
// the pointwise tail of the fused stage as the device launches it: pressure pass (awfl_ptail_kernel), then the tracers' fix-up
// pass, one wavefront per (tracer, x line, member block) (awfl_trfix_kernel)
template <int STAGE>
static void tail_launch(Emu *h, const double *in, const double *p0, double *out, double dt) {
  const Params &P = h->P;
  if (!(h->xtile && h->tile_pressure))
    for (long long idx = 0; idx < P.ncell; idx++) pressure_tail_body(P, out, cell_of(P, idx));
  if (h->xtile && h->tile_pressure && P.nt > 1) return;      // (done by the last slice of the phase-2 tile launch)
  const FctRows rows = fct_rows(h);
  bool any = false;
  for (int b = 0; b < ((P.nens + 63) >> 6); b++) any = any || rows.any[b] == rows.seq;
  if (!any) return;
  if (h->flat) {      // awfl_trfix_flat_kernel: a lane per cell
    for (long long idx = 0; idx < P.ncell; idx++)
      tracer_fixup_cell_body<STAGE>(P, in, p0, out, h->fx.data(), h->fy.data(), h->fz.data(), h->mult.data(), rows, h->seed.data(), dt, P.idWV, cell_of(P, idx));
    return;
  }
  for (int k = 0; k < P.nz; k++)       // water vapour only: the others were completed by phase 2 of their sweeps
    for (int j = 0; j < P.ny; j++)
      for (int e = 0; e < P.nens; e++)
        tracer_fixup_line_body<STAGE>(P, in, p0, out, h->fx.data(), h->fy.data(), h->fz.data(), h->mult.data(), rows, h->seed.data(), dt, P.idWV, k, j, e);
}

static TracerPtrs tptrs(const Emu *h, double *tracers) {
  TracerPtrs tp;
  for (int t = 0; t < MAXT; t++) tp.p[t] = nullptr;
  for (int t = 0; t < h->P.nt; t++) tp.p[t] = tracers + (long long)t * h->P.ncell;
  return tp;
}

extern "C" {

Emu *emu_init(int nens, int nx, int ny, int nz, int nt, double xlen, double ylen, const double *consts6, int idWV,
              const unsigned char *pos, const unsigned char *mass, const double *dz, int seg) {
  Emu *h = new Emu();
  Params &P = h->P;
  std::memset(&P, 0, sizeof(P));
  double R_d = consts6 ? consts6[0] : 287., cp_d = consts6 ? consts6[1] : 1003., R_v = consts6 ? consts6[2] : 461.;
  double p0 = consts6 ? consts6[4] : 1.e5, grav = consts6 ? consts6[5] : 9.81;
  double cv_d = cp_d - R_d, gamma = cp_d / cv_d, kappa = R_d / cp_d;
  P.nens = nens; P.nx = nx; P.ny = ny; P.nz = nz; P.nt = nt; P.sim2d = ny == 1; P.grav_balance = 1; P.seg = seg;
  P.dx = xlen / nx; P.dy = ylen / ny; P.rdx = 1 / P.dx; P.rdy = 1 / P.dy;
  P.C0 = std::pow(R_d * std::pow(p0, -kappa), gamma); P.gamma = gamma; P.grav = grav; P.R_d = R_d; P.R_v = R_v;
  P.sx = nens; P.sy = (long long)nx * nens; P.sz = (long long)ny * nx * nens;
  P.prim_fs = (long long)(nz + 6) * P.sz; P.ncell = (long long)nz * P.sz; P.fz_fs = (long long)(nz + 1) * P.sz;
  P.idWV = idWV;
  for (int t = 0; t < nt; t++) {
    if (pos[t]) P.pos_mask |= 1ull << t;
    if (mass[t]) P.mass_mask |= 1ull << t;
  }
  h->dz.assign(dz, dz + (size_t)nz * nens);
  h->vt = build_vertical_tables(h->dz.data(), nz, nens);
  P.vz_per_ens = h->vt.per_ens;
  h->vz = h->vt.table;
  const double nan = NAN;
  h->prim0.assign((size_t)(6 + nt) * P.prim_fs, nan); h->prim1.assign((size_t)(6 + nt) * P.prim_fs, nan);
  h->prim2.assign((size_t)(6 + nt) * P.prim_fs, nan);
  h->fx.assign((size_t)(5 + nt) * P.ncell, nan); h->fy.assign((size_t)(5 + nt) * P.ncell, nan);
  h->fz.assign((size_t)(5 + nt) * P.fz_fs, nan);
  h->seed.assign((size_t)nt * P.ncell, nan); h->mult.assign((size_t)nt * P.ncell, nan);
  h->fct_flags.assign((size_t)nt * fct_rows_per_tracer(P) + (size_t)((nens + 63) >> 6) + (size_t)nt * fct_lines_per_tracer(P), 0);   // rows + "any" words + lines
  h->grav_var.assign((size_t)nz * nens, nan); h->hy_dens.assign((size_t)nz * nens, nan); h->hy_pres.assign((size_t)nz * nens, nan);
  P.dz = h->dz.data(); P.grav_var = h->grav_var.data(); P.hy_dens = h->hy_dens.data(); P.hy_pres = h->hy_pres.data();
  P.vz = h->vz.data();
  h->rdz.resize((size_t)nz * nens);
  for (size_t i = 0; i < h->rdz.size(); i++) h->rdz[i] = fast_rcp(h->dz[i]);
  P.rdz = h->rdz.data();
  build_pow_tab(h->pow_tab);
  P.pw = &h->pow_tab;
  return h;
}

void emu_destroy(Emu *h) { delete h; }
// pow_pos_fast on the host: the same IEEE operations as on the device (tests/test_pow_pos.py)
void emu_pow(const double *x, int n, double y, double *out) {
  PowTab T;
  build_pow_tab(T);
  for (int i = 0; i < n; i++) out[i] = pow_pos_fast(x[i], y, &T);
}
void emu_set_grav_balance(Emu *h, int v) { h->P.grav_balance = v ? 1 : 0; }
void emu_set_seg(Emu *h, int seg) { h->P.seg = seg; }
void emu_set_span(Emu *h, int span) { h->span = span; }
void emu_set_fused(Emu *h, int fused) { h->fused = fused; }
// the y differences of the state folded into the z sweep's output (3-D, member lanes + sweep kernels; resolve_lane_mapping)
void emu_set_yz_fold(Emu *h, int on) { h->P.yz_fold = (on && !h->P.sim2d && !h->flat && !h->xtile) ? 1 : 0; }
void emu_set_xtr_split(Emu *h, int split) { h->xtr_split = split; }
void emu_set_lane_mapping(Emu *h, int flat, int xtile) { h->flat = flat; h->xtile = xtile; }
void emu_set_x_tile(Emu *h, int w, int tc, int lpb) { h->xt_w = w; h->xt_tc = tc; h->xt_lpb = lpb; }
void emu_set_tile_pressure(Emu *h, int on) { h->tile_pressure = on; }
void emu_set_flux_tile(Emu *h, int on, int tc_y, int tc_z) { h->ftile = on; h->ft_tc_y = tc_y; h->ft_tc_z = tc_z; }
void emu_x_tile_geometry(Emu *h, int *g) {
  const XTileGeom G = xtile_geometry(h->P, h->xt_w, h->xt_tc, h->xt_lpb);
  g[0] = G.W; g[1] = G.nmb; g[2] = G.tc; g[3] = G.halo; g[4] = G.ntl; g[5] = G.lpb;
}
int emu_vz_per_ens(Emu *h) { return h->P.vz_per_ens; }
double *emu_buffer(Emu *h, const char *name) {
  std::string k(name);
  if (k == "prim0") return h->prim0.data();
  if (k == "prim1") return h->prim1.data();
  if (k == "flux_x") return h->fx.data();
  if (k == "flux_y") return h->fy.data();
  if (k == "flux_z") return h->fz.data();
  if (k == "seed") return h->seed.data();
  if (k == "mult") return h->mult.data();
  if (k == "variable_gravity") return h->grav_var.data();
  if (k == "hy_dens_cells") return h->hy_dens.data();
  if (k == "hy_pressure_cells") return h->hy_pres.data();
  if (k == "vert_sten_to_coefs") return h->vt.s2c.data();
  if (k == "vert_weno_recon_lower") return h->vt.wrl.data();
  return nullptr;
}

static void init_prim(Emu *h, double *rho_d, double *u, double *v, double *w, double *T, double *tracers,
                      const double *const *gcm, bool subtract_hy) {
  TracerPtrs tp = tptrs(h, tracers);
  for (long long idx = 0; idx < h->P.ncell; idx++)
    init_prim_body(h->P, rho_d, u, v, w, T, tp, gcm, h->prim0.data(), h->seed.data(), subtract_hy, cell_of(h->P, idx));
}

void emu_declare_hydrostatic(Emu *h, double *rho_d, double *u, double *v, double *w, double *T, double *tracers,
                             const double *const *gcm) {
  init_prim(h, rho_d, u, v, w, T, tracers, gcm, false);
  for (int k = 0; k < h->P.nz; k++)
    for (int e = 0; e < h->P.nens; e++) {
      if (h->P.vz_per_ens) hydro_mean_body<true>(h->P, h->prim0.data(), h->grav_var.data(), h->hy_dens.data(), h->hy_pres.data(), k, e);
      else hydro_mean_body<false>(h->P, h->prim0.data(), h->grav_var.data(), h->hy_dens.data(), h->hy_pres.data(), k, e);
    }
  if (h->P.grav_balance) {
    // what the device does in mode A (awfl_hydro_pint_kernel + awfl_hydro_sum_kernel): the interface pressure of every face first,
    // then the means -- must reproduce hydro_mean_body's variable_gravity bit for bit
    const Params &P = h->P;
    std::vector<double> pint((size_t)P.fz_fs), g2((size_t)P.nz * P.nens);
    for (int kf = 0; kf <= P.nz; kf++)
      for (int j = 0; j < P.ny; j++)
        for (int i = 0; i < P.nx; i++)
          for (int e = 0; e < P.nens; e++) {
            if (P.vz_per_ens) hydro_pint_face<true>(P, h->prim0.data(), pint.data(), kf, j, i, e);
            else hydro_pint_face<false>(P, h->prim0.data(), pint.data(), kf, j, i, e);
          }
    for (int k = 0; k < P.nz; k++)
      for (int e = 0; e < P.nens; e++) hydro_mean_from_pint(P, h->prim0.data(), pint.data(), g2.data(), k, e);
    if (std::memcmp(g2.data(), h->grav_var.data(), g2.size() * sizeof(double)) != 0) abort();
  }
}

double emu_compute_time_step(Emu *h, double *rho_d, double *u, double *v, double *w, double *T, double *tracers, double cfl) {
  double m = INFINITY;
  for (long long idx = 0; idx < h->P.ncell; idx++)
    m = std::fmin(m, cfl_body(h->P, rho_d, u, v, w, T, tracers + (long long)h->P.idWV * h->P.ncell, cfl, idx));
  return m;
}

void emu_convert_coupler_to_dynamics(Emu *h, double *rho_d, double *u, double *v, double *w, double *T, double *tracers) {
  init_prim(h, rho_d, u, v, w, T, tracers, nullptr, !h->P.grav_balance);
}

void emu_flux_stage(Emu *h, double dt) {
  h->fct_seq++;
  flux_launch(h, h->prim0.data());
  fct_launch(h, dt);
}

int emu_time_step(Emu *h, double *rho_d, double *u, double *v, double *w, double *T, double *tracers, double crm_dt,
                  double dt_hint, double *dt_out) {
  emu_convert_coupler_to_dynamics(h, rho_d, u, v, w, T, tracers);
  double dt = dt_hint > 0 ? dt_hint : emu_compute_time_step(h, rho_d, u, v, w, T, tracers, 0.8);
  int ncycles = (int)std::ceil(crm_dt / dt);
  dt = crm_dt / ncycles;
  if (dt_out) *dt_out = dt;
  double *p0 = h->prim0.data(), *p1 = h->prim1.data();
  if (h->fused) {   // as pam_amd_awfl_time_step: three rotating buffers A -> B -> C -> B, then B is the state
    double *A = h->prim0.data(), *B = h->prim1.data(), *C = h->prim2.data();
    // poison the x fluxes of the state: the fused stage must not read them
    for (int ic = 0; ic < ncycles; ic++) {
      std::fill(h->fx.begin(), h->fx.begin() + 5 * h->P.ncell, NAN);   // (field 0 is re-used as the x-sweep's own scratch)
      h->fct_seq++; flux_launch(h, A, 6, true); xupd_launch<1>(h, A, A, B, dt, dt); tail_launch<1>(h, A, A, B, dt);
      h->fct_seq++; flux_launch(h, B, 6, true); xupd_launch<2>(h, B, A, C, dt, (1.0 / 4.0) * dt); tail_launch<2>(h, B, A, C, dt);
      h->fct_seq++; flux_launch(h, C, 6, true); xupd_launch<3>(h, C, A, B, dt, (2.0 / 3.0) * dt); tail_launch<3>(h, C, A, B, dt);
      std::swap(A, B);
    }
    if (A != h->prim0.data()) h->prim0.swap(h->prim1);   // an odd number of sub-steps: the state sits in prim1
    p0 = h->prim0.data();
  } else {
    for (int ic = 0; ic < ncycles; ic++) {
      h->fct_seq++; flux_launch(h, p0); fct_launch(h, dt); update_launch<1>(h, p0, p0, p1, dt);
      h->fct_seq++; flux_launch(h, p1); fct_launch(h, (1.0 / 4.0) * dt); update_launch<2>(h, p1, p0, p1, dt);
      h->fct_seq++; flux_launch(h, p1); fct_launch(h, (2.0 / 3.0) * dt); update_launch<3>(h, p1, p0, p0, dt);
    }
  }
  TracerPtrs tp = tptrs(h, tracers);
  for (long long idx = 0; idx < h->P.ncell; idx++) finalize_body(h->P, p0, h->seed.data(), rho_d, u, v, w, T, tp, cell_of(h->P, idx));
  return ncycles;
}

}  // extern "C"
