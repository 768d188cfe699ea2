// modules/broadcast_initial_gcm_column.h -- modules::broadcast_initial_gcm_column(coupler) with the reference's signature
// (pam_core/modules/broadcast_initial_gcm_column.h:8-41: the GCM columns of density_dry, uvel, vvel, wvel, temp and water_vapor
// copied to every CRM column of their member), forwarding to the C ABI (include/pam_amd_modules.h).
#pragma once
#include <array>

#include "pam_coupler.h"
#include "pam_amd_awfl.h"
#include "pam_amd_modules.h"

namespace modules {

inline void broadcast_initial_gcm_column(pam::PamCoupler &coupler) {
  int nz = coupler.get_nz(), ny = coupler.get_ny(), nx = coupler.get_nx(), nens = coupler.get_nens();
  auto &dm = coupler.get_data_manager_device_readwrite();
  std::array<const double *, 6> gcm;
  std::array<double *, 6> crm;
  char const *gn[6] = {"gcm_density_dry", "gcm_uvel", "gcm_vvel", "gcm_wvel", "gcm_temp", "gcm_water_vapor"};
  char const *cn[6] = {"density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor"};
  for (int f = 0; f < 6; f++) {
    gcm[f] = dm.get<real const, 2>(gn[f]).data();
    crm[f] = dm.get<real, 4>(cn[f]).data();
  }
  if (pam_amd_broadcast_initial_gcm_column(nens, nx, ny, nz, 6, gcm.data(), crm.data(), nullptr)) endrun(pam_amd_awfl_last_error());
}

}  // namespace modules
