// modules_kernels.hip -- gfx950 kernels of the coupler modules around the dycore (include/pam_amd_modules.h).
// sponge_layer: pam_core/modules/sponge_layer.h:8-95.  Both kernels are tiny and HBM-bound (top 5 of 60 levels).
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/pam_amd_awfl.h"
#include "../../include/pam_amd_modules.h"

namespace {

constexpr int MAX_FIELDS = 55;   // 5 state fields + pam_const.h:24 max_fields tracers
struct FieldPtrs { double *p[MAX_FIELDS]; };

// horizontal mean of level k = nz-1-kloc for (field, member), accumulated in the reference's serial atomicAdd order
// (j outer, i inner: sponge_layer.h:73-76) -> deterministic.  wvel (field 3) keeps a zero mean (:34,:75).
__global__ void __launch_bounds__(64) sponge_mean_kernel(FieldPtrs F, int nens, int nx, int ny, int nz, int num_fields,
                                                         int num_layers, double *__restrict__ havg) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)num_fields * num_layers * nens) return;
  const int e = (int)(t % nens);
  const int kloc = (int)((t / nens) % num_layers);
  const int ifld = (int)(t / ((long long)nens * num_layers));
  const int k = nz - 1 - kloc;
  double s = 0.0;
  if (ifld != 3) {
    const double r_nx_ny = 1.0 / (nx * ny);
    const double *f = F.p[ifld] + (long long)k * ny * nx * nens + e;
    for (int j = 0; j < ny; j++)
      for (int i = 0; i < nx; i++) s += f[((long long)j * nx + i) * nens] * r_nx_ny;
  }
  havg[t] = s;
}

// sponge_layer.h:87-93
__global__ void __launch_bounds__(256) sponge_relax_kernel(FieldPtrs F, int nens, int nx, int ny, int nz, int num_fields,
                                                           int num_layers, const double *__restrict__ havg,
                                                           const double *__restrict__ zint, const double *__restrict__ zmid,
                                                           double time_factor) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long per_layer = (long long)ny * nx * nens;
  if (t >= (long long)num_fields * num_layers * per_layer) return;
  const long long c = t % per_layer;
  const int e = (int)(c % nens);
  const int kloc = (int)((t / per_layer) % num_layers);
  const int ifld = (int)(t / (per_layer * num_layers));
  const int k = nz - 1 - kloc;
  const double ztop = zint[(long long)nz * nens + e];
  const double rel_dist = (ztop - zmid[(long long)k * nens + e]) / (ztop - zmid[(long long)(nz - 1 - (num_layers - 1)) * nens + e]);
  const double space_factor = (cos(M_PI * rel_dist) + 1) / 2;
  const double factor = space_factor * time_factor;
  double *f = F.p[ifld] + (long long)k * per_layer + c;
  const double h = havg[((long long)ifld * num_layers + kloc) * nens + e];
  *f += (h - *f) * factor;
}

}  // namespace

extern "C" int pam_amd_set_last_error_(int code, const char *msg);   // defined in awfl_kernels.hip

extern "C" int pam_amd_sponge_layer(int nens, int nx, int ny, int nz, int num_fields, double *const *fields,
                                    const double *zint, const double *zmid, double crm_dt, int num_layers, double time_scale,
                                    double *workspace, void *stream) {
  if (nens < 1 || nx < 1 || ny < 1 || nz < 1 || !fields || !zint || !zmid || !workspace)
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
  const long long n1 = (long long)num_fields * num_layers * nens;
  hipLaunchKernelGGL(sponge_mean_kernel, dim3((unsigned)((n1 + 63) / 64)), dim3(64), 0, s, F, nens, nx, ny, nz, num_fields,
                     num_layers, workspace);
  const long long n2 = n1 * ny * nx;
  hipLaunchKernelGGL(sponge_relax_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, F, nens, nx, ny, nz, num_fields,
                     num_layers, workspace, zint, zmid, crm_dt / time_scale);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return pam_amd_set_last_error_(PAM_AMD_ENOGPU, hipGetErrorString(err));
  return PAM_AMD_OK;
}
