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

#ifdef __cplusplus
}
#endif
#endif
