// modules/perturb_temperature.h -- modules::perturb_temperature(coupler, id, magnitude) with the reference's signature
// (pam_core/modules/perturb_temperature.h:10-63), forwarding to the C ABI (include/pam_amd_modules.h).  The reference draws from
// yakl::Random, which is not in its tree (YAKL is an absent submodule): splitmix64 of the reference's own seed formula stands in --
// everything else (the seed, the range, the decay with height, the rescaling to the unperturbed level mean) follows the reference.
#pragma once
#include "pam_coupler.h"
#include "pam_amd_awfl.h"
#include "pam_amd_modules.h"

namespace modules {

inline void perturb_temperature(pam::PamCoupler &coupler, intConst1d id, real magnitude = 0.1) {
  int nz = coupler.get_nz(), ny = coupler.get_ny(), nx = coupler.get_nx(), nens = coupler.get_nens();
  if ((int)id.size() != nens) endrun("ERROR: size of id array must be the same as nens");        // perturb_temperature.h:20
  auto &dm = coupler.get_data_manager_device_readwrite();
  if (pam_amd_perturb_temperature(nens, nx, ny, nz, dm.get<real, 4>("temp").data(), id.data(), magnitude, nullptr))
    endrun(pam_amd_awfl_last_error());
}

}  // namespace modules
