/*
 * pam_amd_modules.h -- C ABI of coupler modules that surround the dycore in the CRM step loop ("next rows" of
 * SURVEY.md section 8f), exported by the same libpam_amd_awfl.so.  Like the dycore entry points they work in place on
 * the coupler's device arrays ((nz,ny,nx,nens), nens fastest) and return 0 / a negative PAM_AMD_E* code with the
 * message in pam_amd_awfl_last_error().
 */
#ifndef PAM_AMD_MODULES_H
#define PAM_AMD_MODULES_H

#ifdef __cplusplus
extern "C" {
#endif

/* modules::sponge_layer(coupler)  (pam_core/modules/sponge_layer.h:8-95; called right after the dycore,
 * standalone/mmf_simplified/driver.cpp:250).  Relaxes the top `num_layers` levels of every state and tracer field
 * towards their horizontal mean (w: towards zero) with strength crm_dt/time_scale x ((cos(pi d)+1)/2).
 *   fields      host array of num_fields DEVICE pointers in the reference's order: density_dry, uvel, vvel, wvel, temp,
 *               then the tracers in registration order (sponge_layer.h:54-62)
 *   zint, zmid  DEVICE "vertical_interface_height" (nz+1,nens), "vertical_midpoint_height" (nz,nens)
 *   num_layers  option "sponge_num_layers" (default 5), time_scale option "sponge_time_scale" (default 60 s)
 *   workspace   unused since ABI 5 (the horizontal means live in the kernel's workgroups); may be NULL.  ABI <= 4: DEVICE scratch
 *               of num_fields*num_layers*nens doubles
 *   stream      hipStream_t (NULL = default stream) */
int pam_amd_sponge_layer(int nens, int nx, int ny, int nz, int num_fields, double *const *fields, const double *zint,
                         const double *zmid, double crm_dt, int num_layers, double time_scale, double *workspace,
                         void *stream);

/* Frees what the module entry points keep per device and process (the 3.5 KB of pow tables the Kessler kernels read, built on the first
 * call on a device: that call allocates and copies synchronously).  Optional; the entry points rebuild them on demand. */
int pam_amd_modules_finalize(void);

/* Microphysics::timeStep(coupler) of the Kessler scheme  (physics/micro/kessler/Microphysics.h:120-268, kessler():346-457;
 * called after the SGS module, standalone/mmf_simplified/driver.cpp:253).  Works in place on the coupler's DEVICE arrays:
 *   rho_v, rho_c, rho_r   tracers "water_vapor", "cloud_liquid", "precip_liquid" (nz,ny,nx,nens)     in/out
 *   rho_dry, temp         "density_dry" (in), "temp" (in/out)
 *   precl                 "precl" (ny,nx,nens): precipitation rate, m of water per second              out
 *   zmid                  "vertical_midpoint_height" (nz,nens)
 *   dt                    option "crm_dt"; R_d, R_v, cp_d, p0: the scheme's constants (Microphysics.h:26-31)
 *   workspace             DEVICE scratch of nz*ny*nx*nens + 1 doubles (old Exner function + the time-step minimum)
 *   rainsplit_hint        > 0: number of sedimentation sub-cycles to use (ensemble shards pass the value derived from the
 *                         GLOBAL minimum of pam_amd_kessler_max_stable_dt, as the reference's minval is global, :389-390);
 *                         <= 0: computed here with one 8-byte read-back, which synchronises `stream`
 *   rainsplit             out (may be NULL): sub-cycles used */
int pam_amd_kessler_time_step(int nens, int nx, int ny, int nz, double *rho_v, double *rho_c, double *rho_r,
                              const double *rho_dry, double *temp, double *precl, const double *zmid, double dt, double R_d,
                              double R_v, double cp_d, double p0, double *workspace, void *stream, int rainsplit_hint,
                              int *rainsplit);

/* The sedimentation time-step limit min(0.8 dz / velqr) of kessler() (:377-390) for the current state, without changing
 * it; rainsplit = ceil(dt / dt_max).  Synchronises `stream`.  workspace as above. */
int pam_amd_kessler_max_stable_dt(int nens, int nx, int ny, int nz, const double *rho_r, const double *rho_dry,
                                  const double *zmid, double dt, double *workspace, void *stream, double *dt_max);

/* modules::compute_gcm_forcing_tendencies(coupler)  (pam_core/modules/gcm_forcing.h:17-210; once per GCM step).
 * Host arrays of DEVICE pointers, in this order:
 *   crm[10]   (nz,ny,nx,nens): density_dry, uvel, vvel, temp, water_vapor, cloud_water, ice, cloud_water_num, ice_num, rain_num
 *   gcm[10]   (nz,nens): gcm_density_dry, gcm_uvel, gcm_vvel, gcm_temp, gcm_water_vapor, gcm_cloud_water, gcm_cloud_ice,
 *             gcm_num_liq, gcm_num_ice, gcm_num_rain
 *   tend[14]  (nz,nens): gcm_forcing_tend_{rho_d,uvel,vvel,temp,qtot,qv,ql,qi,rho_v,rho_l,rho_i,nc,ni,nr}
 * Writes every tend entry except rho_v, rho_l, rho_i (those are diagnostics of the apply step).  Horizontal means are summed
 * deterministically (strips of cells in the reference's serial order, strips added in ascending order; the reference uses
 * atomicAdd in no particular order); stream-ordered scratch for the strips' partial sums is taken with hipMallocAsync. */
int pam_amd_gcm_forcing_compute(int nens, int nx, int ny, int nz, const double *const *crm, const double *const *gcm,
                                double *const *tend, double gcm_physics_dt, void *stream);

/* modules::apply_gcm_forcing_tendencies(coupler)  (gcm_forcing.h:297-440, fill_holes :213-284; every CRM step): adds
 * tend*crm_dt to the CRM state, clips number concentrations, diagnoses tend rho_v/rho_l/rho_i, and fills negative water
 * with the reference's multiplicative hole filler (per level; over the whole CRM when a level lacks the mass).
 *   dz          DEVICE "vertical_cell_dz" (nz,nens)
 *   workspace   DEVICE scratch of 6*nz*nens + 2*nens + 4 doubles
 *   mask        out (may be NULL): bit s (0 vapour, 1 liquid, 2 ice) = hole filling ran, bit 4+s = its whole-CRM pass ran
 * Synchronises `stream` once (the reference's host reads of sum(neg_mass) and neg_too_large). */
int pam_amd_gcm_forcing_apply(int nens, int nx, int ny, int nz, double *const *crm, const double *const *gcm,
                              double *const *tend, const double *dz, double crm_dt, double gcm_physics_dt, double *workspace,
                              void *stream, int *mask);

/* modules::broadcast_initial_gcm_column(coupler)  (pam_core/modules/broadcast_initial_gcm_column.h:8-41): num_fields = 6,
 * gcm[] = DEVICE (nz,nens) gcm_density_dry, gcm_uvel, gcm_vvel, gcm_wvel, gcm_temp, gcm_water_vapor copied to every column of
 * crm[] = DEVICE (nz,ny,nx,nens) density_dry, uvel, vvel, wvel, temp, water_vapor.  num_fields = 1 is
 * broadcast_initial_gcm_column_dry_density (:44-62). */
int pam_amd_broadcast_initial_gcm_column(int nens, int nx, int ny, int nz, int num_fields, const double *const *gcm,
                                         double *const *crm, void *stream);

/* modules::perturb_temperature(coupler, id, magnitude)  (pam_core/modules/perturb_temperature.h:10-63): random
 * perturbation of "temp" in the lowest nz/4 levels, decaying linearly with height, rescaled per level to the
 * unperturbed horizontal mean.  id: DEVICE int[nens], one stream id per member.  NOT bit-comparable with the reference:
 * its generator is yakl::Random (third-party, absent from the reference tree); splitmix64 of the reference's seed
 * formula is used instead -- everything else (seed, range, decay, rescale, summation order) follows the reference. */
int pam_amd_perturb_temperature(int nens, int nx, int ny, int nz, double *temp, const int *id, double magnitude, void *stream);

/* supercell_init(vert_interface, rho_d_col, uvel_col, vvel_col, wvel_col, temp_col, rho_v_col, Rd, Rv, grav)
 * (standalone/mmf_simplified/supercell_init.h:7-135): the standalone driver's idealised supercell column, which the driver
 * then broadcasts to every CRM cell (pam_amd_broadcast_initial_gcm_column) and perturbs.  vert_interface: DEVICE, nz+1
 * interface heights of ONE column; the six outputs: DEVICE, nz values each. */
int pam_amd_supercell_init(int nz, const double *vert_interface, double R_d, double R_v, double grav, double *rho_d_col,
                           double *uvel_col, double *vvel_col, double *wvel_col, double *temp_col, double *rho_v_col,
                           void *stream);

#ifdef __cplusplus
}
#endif
#endif
