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
 *   workspace   DEVICE scratch of num_fields*num_layers*nens doubles (the horizontal means)
 *   stream      hipStream_t (NULL = default stream) */
int pam_amd_sponge_layer(int nens, int nx, int ny, int nz, int num_fields, double *const *fields, const double *zint,
                         const double *zmid, double crm_dt, int num_layers, double time_scale, double *workspace,
                         void *stream);

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

#ifdef __cplusplus
}
#endif
#endif
