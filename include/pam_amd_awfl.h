/*
 * pam_amd_awfl.h -- C ABI of the MI355X-native AWFL dynamical-core step (libpam_amd_awfl.so).
 *
 * The reference plug-in boundary for this path is the compile-time duck-typed C++ class `Dycore`
 * (E3SM-Project/PAM dynamics/awfl/Dycore.h) called by the host model through a `pam::PamCoupler`.
 * Each entry point below replaces one member of that class; the thin C++ `class Dycore` a PAM maintainer
 * drops into dynamics/awfl_amd/  forwards to these with the raw device pointers it gets
 * from the coupler's DataManager.  Plain pointers and sizes only; every array is fp64 (`typedef double real`,
 * pam_core/pam_const.h:22) on the device, row-major with nens fastest, exactly as the coupler stores it
 * (pam_core/pam_coupler.h:259-266).
 *
 * Error convention: the reference's `endrun()` prints to stderr and throws (pam_core/pam_const.h:249-252).
 * Here every function returns 0 on success or a negative PAM_AMD_E* code; pam_amd_awfl_last_error() returns
 * the message (thread-local).  The C++ adaptor turns a non-zero return into `endrun(message)`.
 *
 * Threading: one handle per coupler, used from one host thread at a time (as pam_interface.h:26-32 keeps
 * one coupler per thread).  All kernels are enqueued on the handle's HIP stream; calls that return a value
 * to the host (compute_time_step, and time_step when it derives dt itself) synchronise that stream.
 */
#ifndef PAM_AMD_AWFL_H
#define PAM_AMD_AWFL_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PAM_AMD_AWFL_ABI_VERSION 5

#define PAM_AMD_OK 0
#define PAM_AMD_EINVAL (-1)   /* bad argument / inconsistent dimensions (reference: endrun) */
#define PAM_AMD_ENOGPU (-2)   /* no HIP device, or a HIP runtime call failed */
#define PAM_AMD_ENOMEM (-3)   /* device allocation failed */
#define PAM_AMD_ESTATE (-4)   /* call sequence violated (e.g. time_step before init) */

typedef struct pam_amd_awfl pam_amd_awfl_t; /* opaque dycore handle */

/* Grid, tracer registry and physical constants the dycore reads from the coupler in Dycore::init
 * (dynamics/awfl/Dycore.h:835-984).  A constant given as NaN is "option absent": the six primary constants then take
 * the defaults of Dycore.h:871-876 and the five derived ones are derived as in Dycore.h:883-890 (each only if absent,
 * from the values in force -- a host model that pre-set e.g. gamma_d or C0 gets exactly its value, as :942-950 reads
 * them back from the coupler). */
typedef struct pam_amd_awfl_config {
  int nens, nx, ny, nz;                 /* coupler.get_nens/nx/ny/nz (pam_coupler.h:70-91) */
  int num_tracers;                      /* coupler.get_num_tracers() (>=1: water_vapor) */
  double xlen, ylen;                    /* coupler.get_xlen/ylen (m) */
  double R_d, cp_d, R_v, cp_v, p0, grav;/* options of the same names (Dycore.h:871-876) */
  double cv_d, gamma_d, kappa_d, cv_v, C0; /* options of the same names (Dycore.h:883-890); NaN = absent = derive */
  int idWV;                             /* index of tracer "water_vapor" (Dycore.h:969,974) */
  const unsigned char *tracer_positive; /* host, num_tracers flags (coupler.get_tracer_info, Dycore.h:963-970) */
  const unsigned char *tracer_adds_mass;/* host, num_tracers flags */
  const double *vertical_cell_dz;       /* DEVICE, (nz,nens): coupler entry "vertical_cell_dz" (Dycore.h:894) */
  void *stream;                         /* hipStream_t to enqueue on (NULL = default stream) */
} pam_amd_awfl_config_t;

/* The coupler fields the dycore reads and writes (Dycore.h:1301-1310, :1357-1366), all DEVICE (nz,ny,nx,nens). */
typedef struct pam_amd_awfl_fields {
  double *density_dry, *uvel, *vvel, *wvel, *temp;
  double *const *tracers;               /* host array of num_tracers device pointers, coupler registration order */
} pam_amd_awfl_fields_t;

/* GCM columns for declare_current_profile_as_hydrostatic(use_gcm_data=true) (Dycore.h:1416-1420), DEVICE (nz,nens) */
typedef struct pam_amd_awfl_gcm_columns {
  const double *gcm_density_dry, *gcm_temp, *gcm_water_vapor, *gcm_cloud_water, *gcm_cloud_ice;
} pam_amd_awfl_gcm_columns_t;

int pam_amd_awfl_abi_version(void);
const char *pam_amd_awfl_last_error(void);

/* Dycore::init (Dycore.h:835).  Allocates every scratch array once (the reference re-allocates ~13 arrays per
 * sub-cycle, Dycore.h:149-152,307,325-330), builds the per-level vertical WENO matrices (Dycore.h:897-940) and
 * sets option balance_hydrostasis_with_gravity = true (Dycore.h:866). */
int pam_amd_awfl_init(const pam_amd_awfl_config_t *cfg, pam_amd_awfl_t **out);

/* The optional idealised initial conditions of Dycore::init (Dycore.h:986-1090, compiled into the reference under
 * PAM_STANDALONE): init_data is the value of the `initData` key of the YAML file named by option
 * "standalone_input_file" -- "thermal" (Dycore.h:1021-1088), "supercell" (init_supercell, Dycore.h:1096-1276) or
 * "external" (no-op); anything else is the reference's endrun("ERROR: Invalid data_spec").  Fills the coupler fields.
 * vertical_midpoint_height (nz,nens) and vertical_interface_height (nz+1,nens) are the coupler entries of those names
 * (DEVICE). */
int pam_amd_awfl_init_idealized(pam_amd_awfl_t *h, const pam_amd_awfl_fields_t *fields, const char *init_data,
                                const double *vertical_midpoint_height, const double *vertical_interface_height);

/* Dycore::finalize (Dycore.h:1548) + release of the handle. */
int pam_amd_awfl_finalize(pam_amd_awfl_t *h);

/* Dycore::dycore_name (Dycore.h:1544). */
const char *pam_amd_awfl_dycore_name(const pam_amd_awfl_t *h);

/* Options the dycore owns or derives (Dycore.h:866-891): "R_d","R_v","cp_d","cp_v","p0","grav","cv_d","cv_v",
 * "gamma_d","kappa_d","C0" (real) and "balance_hydrostasis_with_gravity","idWV" (returned as 0/1 resp. index). */
int pam_amd_awfl_get_option(const pam_amd_awfl_t *h, const char *key, double *value);
/* coupler.set_option<bool>("balance_hydrostasis_with_gravity", v) after init() selects mode B. */
int pam_amd_awfl_set_balance_hydrostasis_with_gravity(pam_amd_awfl_t *h, int value);

/* Dycore-owned DataManager entries (Dycore.h:868,897-898,983-984): "variable_gravity", "hy_dens_cells",
 * "hy_pressure_cells" (nz,nens); "vert_sten_to_coefs" (nz+2,5,5,nens); "vert_weno_recon_lower" (nz+2,3,3,3,nens).
 * Returns the DEVICE pointer and the dimensions so the host model can register them in its DataManager. */
int pam_amd_awfl_get_array(pam_amd_awfl_t *h, const char *name, double **device_ptr, int dims[5], int *ndims);
/* The reference's DataManager OWNS those entries (dm.register_and_allocate, Dycore.h:868,897-898,983-984) and they
 * outlive dycore.finalize().  bind_array makes the kernels read and write `device_ptr` (storage the host model's
 * DataManager allocated with the dims get_array reports) instead of the handle's own buffer; the current contents
 * are carried over.  Unbound arrays stay in handle-owned storage that pam_amd_awfl_finalize releases. */
int pam_amd_awfl_bind_array(pam_amd_awfl_t *h, const char *name, double *device_ptr);

/* Dycore::declare_current_profile_as_hydrostatic(coupler, use_gcm_data) (Dycore.h:1392).  gcm == NULL is
 * use_gcm_data=false. */
int pam_amd_awfl_declare_current_profile_as_hydrostatic(pam_amd_awfl_t *h, const pam_amd_awfl_fields_t *fields,
                                                        const pam_amd_awfl_gcm_columns_t *gcm);

/* Dycore::compute_time_step(coupler, cfl) (Dycore.h:65): min over every cell and ensemble member held by THIS
 * handle.  When the ensemble is sharded over several GPUs the caller min-reduces the shard values (8 bytes)
 * and passes the result to time_step as dt_dyn_hint, so that all shards sub-cycle identically (Dycore.h:141-145). */
int pam_amd_awfl_compute_time_step(pam_amd_awfl_t *h, const pam_amd_awfl_fields_t *fields, double cfl, double *dt);

/* Dycore::timeStep(coupler) (Dycore.h:107).  crm_dt = option "crm_dt" (Dycore.h:120).  dt_dyn_hint <= 0: derive
 * the dynamics step from compute_time_step(cfl=0.8) on this handle's members (one stream synchronisation);
 * > 0: use it (no host synchronisation).  Outputs (may be NULL): number of sub-cycles and the dt used. */
int pam_amd_awfl_time_step(pam_amd_awfl_t *h, const pam_amd_awfl_fields_t *fields, double crm_dt, double dt_dyn_hint,
                           int *ncycles, double *dt_dyn);

/* Dycore::convert_coupler_to_dynamics / convert_dynamics_to_coupler (Dycore.h:1336, :1281) on the dycore's
 * resident state (density-divided form). */
int pam_amd_awfl_convert_coupler_to_dynamics(pam_amd_awfl_t *h, const pam_amd_awfl_fields_t *fields);
int pam_amd_awfl_convert_dynamics_to_coupler(pam_amd_awfl_t *h, const pam_amd_awfl_fields_t *fields);

/* The same two members with the reference's own argument lists (Dycore.h:1336-1338, :1281-1283): the caller's HALO'D device
 * arrays  state(5, nz+6, ny+6, nx+6, nens) = rho, rho u, rho v, rho w, rho theta  and  tracers(NT, nz+6, ny+6, nx+6, nens) =
 * tracer densities, interior at [3+k][3+j][3+i].  coupler -> arrays fills the interior (halos untouched, Dycore.h:1370-1387);
 * arrays -> coupler reads it (Dycore.h:1313-1330).  The handle's resident state is not involved. */
int pam_amd_awfl_convert_coupler_to_dynamics_arrays(pam_amd_awfl_t *h, const pam_amd_awfl_fields_t *fields, double *state,
                                                    double *tracers);
int pam_amd_awfl_convert_dynamics_to_coupler_arrays(pam_amd_awfl_t *h, const pam_amd_awfl_fields_t *fields, const double *state,
                                                    const double *tracers);

/* --- measurement hooks (not in the reference; cf. its -DPAM_FUNCTION_TIMERS, pam_coupler.h:144-150) ------------ */
/* Enable HIP-event timing of every kernel launch on the handle's stream (off by default: zero overhead). */
int pam_amd_awfl_set_kernel_timing(pam_amd_awfl_t *h, int enable);
/* Accumulated device time (ms) and launch count of kernel `name` ("flux","fct_mult","update","init_prim",
 * "finalize","cfl","hydro","xupd","xtr1","xtr2","ptail","trfix") since the last reset; synchronises the stream. */
int pam_amd_awfl_get_kernel_timing(pam_amd_awfl_t *h, const char *name, double *total_ms, long long *launches);
int pam_amd_awfl_reset_kernel_timing(pam_amd_awfl_t *h);
/* The reference's runtime self-check (Dycore.h:36-58 compute_mass; under -DPAM_DEBUG: :136-138 before, :224-251 after the sub-steps of a
 * timeStep) as an opt-in: with enable = 1 every timeStep forms, per variable (tracer densities in registration order, then rho, then
 * rho*theta) and per member, the mean of q * dz over the member's cells before and after, by a device reduction, and synchronises the
 * handle's stream at its end.  A (variable, member) pair whose mass changed by more than 1e-10 relative AND 1e-10 absolute -- the
 * reference's WARNING condition -- counts as a violation; get_conservation returns their number, the largest relative change and where
 * it was; conservation_report the reference's WARNING lines (the first 32).  Off by default: nothing is launched, nothing synchronised.
 * (ABI 5) */
int pam_amd_awfl_set_debug_conservation(pam_amd_awfl_t *h, int enable);
int pam_amd_awfl_get_conservation(pam_amd_awfl_t *h, int *violations, double *max_rel_diff, int *worst_variable, int *worst_member);
const char *pam_amd_awfl_conservation_report(const pam_amd_awfl_t *h);
/* Test hook of the graph replay's error path: the NEXT capture reports a failure of hipStreamBeginCapture (1), hipStreamEndCapture (2) or
 * hipGraphInstantiate (3); the step must then run eagerly, switch the replay off for the handle and leave a "warning: ..." text in
 * pam_amd_awfl_last_error() while returning PAM_AMD_OK. */
int pam_amd_awfl_debug_fail_next_capture(pam_amd_awfl_t *h, int which);
/* Test hook of the check: in the NEXT timeStep, between the last stage and the final masses, one cell of `variable` of `member` is
 * multiplied by `factor` (rho: at constant rho*theta). */
int pam_amd_awfl_debug_inject_mass_fault(pam_amd_awfl_t *h, int variable, int k, int j, int i, int member, double factor);
/* Sweep-kernel tuning knobs; results do not depend on them (bit for bit).
 *   segment: shortest span (faces) a line may be cut into when the ensemble alone does not fill the chip (default 8; 1..64)
 *   span:    faces swept by one wavefront (0 = automatic: the whole line, cut only for small ensembles; at most 64 per span) */
int pam_amd_awfl_set_flux_segment(pam_amd_awfl_t *h, int faces);
int pam_amd_awfl_set_flux_span(pam_amd_awfl_t *h, int faces);
/* Ensemble ranges inside one handle: the members are split into `chunks` contiguous ranges advanced on internal HIP
 * streams (forked from / joined to the handle's stream with events).  chunks = 0: automatic -- two independent ranges for the
 * fused stage from 128 members on, one below; 1-16 for the three-kernel stage, whose HBM-bound
 * update kernel overlaps the FP64-bound flux kernel of another range.  flux_lds_floor_bytes: dynamic LDS requested per flux
 * workgroup when chunks > 1 (the kernel itself uses none: it caps the flux kernel's residency per CU so that another range's
 * blocks can co-reside; default 0 = no cap).  Results do not depend on either.
 * The internal streams belong to the DEVICE, not to the handle: the k-th range of every handle created on a device uses the same
 * stream, created on first use and kept for the life of the process (the runtime maps streams onto a few hardware queues in creation
 * order; a second set of range streams in a process left the ranges of every later handle on one queue).  Handles driven from several
 * host threads at once are therefore ordered on those streams in the order of their calls; results are unaffected. */
int pam_amd_awfl_set_ensemble_chunks(pam_amd_awfl_t *h, int chunks, int flux_lds_floor_bytes);
/* Fused stage with several member ranges: independent = 1 (default) runs every range's whole stage on the range's own stream (no
 * shared compute stream, no events between ranges), so that launches which do not fill the chip overlap their ramp-up and drain
 * phases with another range's kernels; 0: the polynomial kernels of all ranges back to back on one stream (round 2's schedule).
 * Same results.  The automatic range count (set_ensemble_chunks(0)) of the fused stage is 2 from 128 members on. */
int pam_amd_awfl_set_range_schedule(pam_amd_awfl_t *h, int independent);
/* Member-lane sweeps with further tracers (NT > 1): the last three launches of a stage -- phase 2 of the further tracers' x sweeps, the
 * pressure pass, water vapour's fix-up; they read what earlier launches wrote and write disjoint things -- as ONE launch (mode 2;
 * automatic: while the phase-2 launch is less than about two rounds of wavefronts, e.g. one GPU's shard of C4) or three (mode 1).
 * Same bits (ABI 5). */
int pam_amd_awfl_set_tail_fusion(pam_amd_awfl_t *h, int mode);
/* Fused stage, 3-D grids swept with member lanes: the momentum components and rho*theta take their divergence as x + (y + z)
 * (Dycore.h:553-571 sums the three directions; the density and the tracers keep (x + y) + z).  mode 2: the z sweep runs in a launch
 * of its own behind the y sweep, reads the y sweep's flux differences and stores the y+z part, ONE field per variable, which is all
 * the fused x-sweep then loads (it is the HBM-bound kernel of the stage); mode 1 / automatic: the z sweep stores its own differences
 * and the x-sweep loads both and forms the same sum (measured: the flux kernel pays back what the x-sweep gains).
 * Same bits either way (ABI 5). */
int pam_amd_awfl_set_yz_fold(pam_amd_awfl_t *h, int mode);

/* Stage structure.  1 (default): per stage  flux(y,z) -> fused x-sweep + update of the state and of the first tracer (incl.
 * its FCT multiplier) -> [FCT multiplier of further tracers] -> pointwise tail (further tracers, the first tracer where the
 * limiter acted, next stage's pressure); the state's x fluxes never reach HBM.  0: flux(x,y,z) ->
 * FCT multiplier -> update, every face flux and multiplier stored.  Both produce the same bits (tests/test_fused_stage.py). */
int pam_amd_awfl_set_fused_stage(pam_amd_awfl_t *h, int enable);

/* Lane mapping of the fused stage; results do not depend on it (bit for bit, tests/test_lane_mapping.py).
 * The coupler stores every field with nens fastest and x next (pam_core/pam_coupler.h:259-263).  Large ensembles run with MEMBER
 * lanes: a wavefront is 64 consecutive members of one grid line and sweeps the line serially (stencil in the lane's registers).
 * Every input file the reference ships has nens = 1 (standalone/mmf_simplified/inputs/input_pama.yaml:14): there a member-lane
 * wavefront is one lane.  Small ensembles therefore run with lanes over the flattened (x, member) axis:
 *   yz_lanes   1: member lanes; 2: FLAT lanes -- a wavefront of a y / z sweep takes 64 consecutive (x, member) pairs (and rows of
 *              them), the stencil stays in the lane;
 *   x_kernels  1: sweep kernels (a wavefront per span of an x line); 2: TILE kernels -- a lane per cell, one polynomial set per
 *              lane, right-edge values and face fluxes handed to the neighbouring lanes (distance nens) through LDS;
 *   0 = automatic for either: flat / tile when nens < 64.
 * set_x_tile tunes the tile geometry (0 = automatic): lanes per row (members of one cell that sit in one row), cells a tile
 * completes (a tile shorter than the line gets one halo row on each side), lines per workgroup (whole-line tiles only). */
int pam_amd_awfl_set_lane_mapping(pam_amd_awfl_t *h, int yz_lanes, int x_kernels);
int pam_amd_awfl_set_x_tile(pam_amd_awfl_t *h, int row_lanes, int cells_per_tile, int lines_per_group);
/* How the lanes of an x tile kernel exchange values with the lanes of the neighbouring cells (the four outer values of each 5-point
 * stencil, the right-edge values of the cell to the left, the fluxes of the right face): mode 1 through an LDS image of the tile and
 * four workgroup barriers; mode 2 by WAVEFRONT SHUFFLES (ds_bpermute_b32 pairs; no LDS, no barrier) -- possible when a whole periodic
 * line of a tile lies inside one wavefront, i.e. nx x (lanes per row) divides 64: the 32-cell lines of the C1 / C2 grid with one or two
 * members, 16-cell lines with up to four; 0 = automatic (shuffles wherever possible).  Same bits either way (ABI 4). */
int pam_amd_awfl_set_x_exchange(pam_amd_awfl_t *h, int mode);
/* With flat y/z lanes the y and z fluxes of a stage can run as ONE tile kernel as well (a lane per cell; rows of a tile follow the
 * sweep direction, the lanes of a row are contiguous (x, member) / (y, x, member) items) instead of flat-lane SWEEPS (a lane per item
 * walks its line serially): enable = 0 automatic (tile kernel while the whole ensemble is below ~2.6e5 cells), 1 sweeps, 2 tile
 * kernel; cells per y tile / levels per z tile, 0 = automatic.  Same bits either way. */
int pam_amd_awfl_set_flux_tile(pam_amd_awfl_t *h, int enable, int cells_per_y_tile, int levels_per_z_tile);
/* The parts of a y/z flux tile -- the acoustic triple (face mass flux + normal momentum) and the groups of three advected quantities --
 * run BEHIND each other in one workgroup (mode 1: one exchange + barrier per part, a serial chain) or BESIDE each other in workgroups
 * of their own (mode 2: the workgroups of the advected groups rebuild the face mass flux from the polynomials of rho*u_n and p
 * instead of waiting for it: same values, same functions, same bits); 0 = automatic.  (ABI 4) */
int pam_amd_awfl_set_flux_tile_parts(pam_amd_awfl_t *h, int mode);
/* x tile kernels of small ensembles, where a launch costs more than its work: the state kernel (awfl_xupd_tile_kernel) also makes the
 * next stage's pressure + density / pressure ghosts (Dycore.h:310-321, :682-709; otherwise awfl_ptail_kernel) and phase 1 of the
 * further tracers (their FCT multipliers, Dycore.h:525-540; otherwise awfl_xtr_tile_kernel<., 1>): two launches less per stage.
 * mode 2 = fused (phase 1 of the tracers BEHIND the state pass of the same wavefront), 3 = fused with phase 1 in workgroups of its own
 * BESIDE the state pass (z slices of the same launch; they rebuild the face mass flux from the polynomials of rho*u and p instead of
 * waiting for it: same values, same functions, same bits), 1 = separate launches, 0 = automatic (fused while the ensemble is below ~1e6
 * cells; beside the state pass while every workgroup of the launch still finds a CU of its own).  Same bits either way. */
int pam_amd_awfl_set_tile_fusion(pam_amd_awfl_t *h, int mode);
/* The state pass of the fused x tile kernel (seven polynomials, face, finish, pressure: one chain per lane) in THREE parts beside each
 * other -- u (+ the stores of the new density and the face mass flux) | v, w | theta, pressure, water vapour -- in workgroups of their
 * own (z slices of the launch), each rebuilding the polynomials of rho*u and p, the face mass flux and the new density instead of
 * waiting for them: same values, same functions, same bits.  mode 2 = parts, 1 = one lane per cell does all, 0 = automatic (parts while
 * every workgroup of the launch can be resident at once).  Implies tracer phase 1 beside the state pass.  (ABI 4) */
int pam_amd_awfl_set_tile_state_parts(pam_amd_awfl_t *h, int mode);
/* Launch-bound ensembles: a whole time_step (coupler -> dycore, every stage, dycore -> coupler: ~10 launches per sub-step of a few
 * microseconds each) is captured once into a HIP graph on an internal stream and replayed -- one graph per (coupler arrays, number
 * of sub-cycles, buffer parity); the caller's stream is ordered before and after it with events.  mode 2 = on (fused stage, one
 * member range), 0 / 1 = off: the DEFAULT, because on MI355X / ROCm 7.2 the replay is 5-15 % slower than the eager launches (the gaps
 * between dependent kernels are the same).  Same launches, same results. */
int pam_amd_awfl_set_graph_replay(pam_amd_awfl_t *h, int mode);
/* Process-wide DEFAULTS of the launch-shape thresholds of the sweep kernels, in wavefronts (experiments; results do not depend on them): a sweep is
 * cut into spans until it has `want_units` wavefronts (> 0; default 3072; the further tracers' own launches: twice that); the y/z sweeps run pass 1 and the field pairs in launches
 * of their own below `two_phase_below` (line, span) units (>= 0; 8192); phase 1 of the further tracers' x sweeps is a launch of its
 * own below `split_below` units (>= 0; 8192).  Negative / zero arguments leave a threshold as it is. */
int pam_amd_awfl_set_launch_tuning(long long want_units, long long two_phase_below, long long split_below);
/* The same three thresholds for ONE handle (ABI 4).  Every handle owns its thresholds: the process-wide call above only sets the
 * defaults a handle created AFTERWARDS starts from, so that in the one-process / N-handle host path (examples/driver.cpp --gpus N)
 * tuning one handle never re-shapes the launches of another.  Drains the handle's streams and rebuilds its member ranges. */
int pam_amd_awfl_set_handle_launch_tuning(pam_amd_awfl_t *h, long long want_units, long long two_phase_below, long long split_below);
/* Separately launched x sweeps of the further tracers (awfl_xtr_kernel / awfl_xtrn_kernel); same bits whatever is chosen (ABI 4).
 *   tracers_per_wavefront   0 = automatic (measured, DESIGN.md section 6: phase 1 one tracer per wavefront -- 83 registers, five
 *                           wavefronts per SIMD; phase 2 pairs -- the densities and the face mass flux are loaded once per pair -- except
 *                           for three further tracers, where three equal wavefronts beat one double and one single); 1, 2, 4 force a
 *                           grouping in both phases (four: half the wavefronts at twice the registers; slower everywhere measured);
 *   prefetch                1: phase 2 requests the loads of the next trip one trip ahead (pairs only: tracers_per_wavefront = 2; one
 *                           more set of loaded values in registers, 2 instead of 3 wavefronts per SIMD; slower, kept as a measured
 *                           experiment). */
int pam_amd_awfl_set_tracer_grouping(pam_amd_awfl_t *h, int tracers_per_wavefront, int prefetch);
/* the resolved mapping: y/z lanes (0 member, 1 flat-lane sweeps, 2 flat lanes + tile kernel), x tile kernels (0 sweeps, 1 tiles with
 * LDS exchange, 2 tiles with wavefront shuffles), pointwise kernels on a
 * grid flat over every cell (0/1 each) and the x tile
 * geometry {lanes per row, member blocks per line, cells per tile, halo rows per side, tiles per line, lines per workgroup} */
int pam_amd_awfl_get_lane_mapping(const pam_amd_awfl_t *h, int *yz_flat, int *x_tiles, int *flat_cells, int geom[6]);

/* --- test hooks: read-only views of resident device buffers, and a single tendency stage ------------------------- */
/* name: "prim0","prim1","prim2","flux_x","flux_y","flux_z","seed","mult". */
int pam_amd_awfl_debug_get_buffer(pam_amd_awfl_t *h, const char *name, double **device_ptr, size_t *nelem);
/* Row flags of the FCT limiter (a row = (tracer, cell, 64 consecutive members)) as the MOST RECENT tendency
 * stage left them: rows in which some member was limited in that stage, all rows, and the "some row was flagged" words OR-ed over
 * the member blocks (fused stage: some row of WATER VAPOUR (idWV) in any block of 64 members -- the words the fix-up pass looks at;
 * three-kernel stage: of any tracer).
 * Synchronises the handle's stream.  Tests use it to prove that a case exercises the limiter's sparse paths. */
int pam_amd_awfl_debug_fct_rows(pam_amd_awfl_t *h, long long *rows_flagged, long long *rows_total, int *any_flagged);
/* The device WENO reconstruction on n stencils of 5 values (DEVICE, (n,5)): left[i]/right[i] = value at the left/right edge
 * of the centre cell (Dycore.h:591-604 with ind = 0/1).  level < 0: the constant uniform-grid matrices (x, y sweeps);
 * 0 <= level <= nz+1: this handle's vertical matrices of that index, member 0 (z sweep, Dycore.h:454-469). */
int pam_amd_awfl_debug_weno(pam_amd_awfl_t *h, int level, const double *stencils, int n, double *left, double *right);
/* The device's x^y for positive x (pow_pos_fast, awfl_device.h: every pow of the step) on n values (DEVICE arrays). */
int pam_amd_awfl_debug_pow(pam_amd_awfl_t *h, const double *x, int n, double y, double *out);
/* ONE tendency stage (stage 1 of a sub-step of length dt_dyn: a forward-Euler step of the resident state, Dycore.h:156-176)
 * with the selected stage structure; afterwards "prim0" holds the new density-divided state. */
int pam_amd_awfl_debug_stage(pam_amd_awfl_t *h, double dt_dyn);
/* flux + FCT multiplier of stage input prim0 with stage time step dt (Dycore.h:334-550); no update. */
int pam_amd_awfl_debug_flux_stage(pam_amd_awfl_t *h, double dt);

#ifdef __cplusplus
}
#endif
#endif
