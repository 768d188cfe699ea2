// modules/sponge_layer.h -- modules::sponge_layer(coupler) with the reference's signature
// (pam_core/modules/sponge_layer.h:8), forwarding to pam_amd_sponge_layer (include/pam_amd_modules.h).
#pragma once
#include <vector>

#include "pam_coupler.h"
#include "pam_amd_awfl.h"
#include "pam_amd_modules.h"

namespace modules {

inline void sponge_layer(pam::PamCoupler &coupler) {
  int nz = coupler.get_nz(), ny = coupler.get_ny(), nx = coupler.get_nx(), nens = coupler.get_nens();
  int num_layers = coupler.option_exists("sponge_num_layers") ? coupler.get_option<int>("sponge_num_layers") : 5;      // :17-18
  real time_scale = coupler.option_exists("sponge_time_scale") ? coupler.get_option<real>("sponge_time_scale") : 60;   // :21-22
  auto &dm = coupler.get_data_manager_device_readwrite();
  std::vector<double *> f;
  for (char const *n : {"density_dry", "uvel", "vvel", "wvel", "temp"}) f.push_back(dm.get<real, 4>(n).data());
  for (auto &n : coupler.get_tracer_names()) f.push_back(dm.get<real, 4>(n).data());
  // (no scratch: the horizontal means live in the kernel's workgroups)
  int rc = pam_amd_sponge_layer(nens, nx, ny, nz, (int)f.size(), f.data(), dm.get<real const, 2>("vertical_interface_height").data(),
                                dm.get<real const, 2>("vertical_midpoint_height").data(), coupler.get_option<real>("crm_dt"), num_layers,
                                time_scale, nullptr, nullptr);
  if (rc) endrun(pam_amd_awfl_last_error());
}

}  // namespace modules
