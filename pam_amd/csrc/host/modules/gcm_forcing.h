// modules/gcm_forcing.h -- modules::compute_gcm_forcing_tendencies(coupler) and modules::apply_gcm_forcing_tendencies(coupler)
// with the reference's signatures (pam_core/modules/gcm_forcing.h:17, :297), forwarding to the C ABI
// (include/pam_amd_modules.h).  Entry names and options ("gcm_physics_dt", "crm_dt") are the reference's.
#pragma once
#include <array>
#include <string>

#include "pam_coupler.h"
#include "pam_amd_awfl.h"
#include "pam_amd_modules.h"

namespace modules {

namespace gcm_forcing_detail {
inline std::array<double *, 10> crm_fields(pam::PamCoupler &coupler) {
  auto &dm = coupler.get_data_manager_device_readwrite();
  std::array<double *, 10> p;
  int n = 0;
  for (char const *name : {"density_dry", "uvel", "vvel", "temp", "water_vapor", "cloud_water", "ice", "cloud_water_num",
                           "ice_num", "rain_num"})
    p[n++] = dm.get<real, 4>(name).data();
  return p;
}
inline std::array<double *, 10> gcm_columns(pam::PamCoupler &coupler) {
  auto &dm = coupler.get_data_manager_device_readwrite();
  std::array<double *, 10> p;
  int n = 0;
  for (char const *name : {"gcm_density_dry", "gcm_uvel", "gcm_vvel", "gcm_temp", "gcm_water_vapor", "gcm_cloud_water",
                           "gcm_cloud_ice", "gcm_num_liq", "gcm_num_ice", "gcm_num_rain"})
    p[n++] = dm.get<real, 2>(name).data();
  return p;
}
inline std::array<double *, 14> tendencies(pam::PamCoupler &coupler) {
  auto &dm = coupler.get_data_manager_device_readwrite();
  std::array<double *, 14> p;
  int n = 0;
  for (char const *name : {"rho_d", "uvel", "vvel", "temp", "qtot", "qv", "ql", "qi", "rho_v", "rho_l", "rho_i", "nc", "ni", "nr"})
    p[n++] = dm.get<real, 2>(std::string("gcm_forcing_tend_") + name).data();
  return p;
}
}  // namespace gcm_forcing_detail

inline void compute_gcm_forcing_tendencies(pam::PamCoupler &coupler) {
  int nz = coupler.get_nz(), ny = coupler.get_ny(), nx = coupler.get_nx(), nens = coupler.get_nens();
  auto &dm = coupler.get_data_manager_device_readwrite();
  if (!dm.entry_exists("gcm_forcing_tend_uvel"))                                         // gcm_forcing.h:132-147
    for (char const *name : {"rho_d", "uvel", "vvel", "temp", "qtot", "qv", "ql", "qi", "rho_v", "rho_l", "rho_i", "nc", "ni", "nr"})
      dm.register_and_allocate<real>(std::string("gcm_forcing_tend_") + name, "GCM forcing", {nz, nens}, {"z", "nens"});
  auto c = gcm_forcing_detail::crm_fields(coupler);
  auto g = gcm_forcing_detail::gcm_columns(coupler);
  auto t = gcm_forcing_detail::tendencies(coupler);
  if (pam_amd_gcm_forcing_compute(nens, nx, ny, nz, c.data(), g.data(), t.data(), coupler.get_option<real>("gcm_physics_dt"), nullptr))
    endrun(pam_amd_awfl_last_error());
}

inline void apply_gcm_forcing_tendencies(pam::PamCoupler &coupler) {
  int nz = coupler.get_nz(), ny = coupler.get_ny(), nx = coupler.get_nx(), nens = coupler.get_nens();
  auto &dm = coupler.get_data_manager_device_readwrite();
  auto c = gcm_forcing_detail::crm_fields(coupler);
  auto g = gcm_forcing_detail::gcm_columns(coupler);
  auto t = gcm_forcing_detail::tendencies(coupler);
  // scratch in a DataManager entry registered on first use (like gcm_forcing_tend_*, gcm_forcing.h:132-147): no allocation and no
  // device-wide synchronisation per CRM step
  int nwork = 6 * nz * nens + 2 * nens + 4;
  if (dm.entry_exists("gcm_forcing_scratch") && (int)dm.get<real, 1>("gcm_forcing_scratch").size() != nwork)
    dm.unregister_and_deallocate("gcm_forcing_scratch");
  if (!dm.entry_exists("gcm_forcing_scratch"))
    dm.register_and_allocate<real>("gcm_forcing_scratch", "GCM forcing column sums", {nwork});
  int rc = pam_amd_gcm_forcing_apply(nens, nx, ny, nz, c.data(), g.data(), t.data(), dm.get<real const, 2>("vertical_cell_dz").data(),
                                     coupler.get_option<real>("crm_dt"), coupler.get_option<real>("gcm_physics_dt"),
                                     dm.get<real, 1>("gcm_forcing_scratch").data(), nullptr, nullptr);
  if (rc) endrun(pam_amd_awfl_last_error());
}

}  // namespace modules
