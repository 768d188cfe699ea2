// physics/micro/kessler_amd/Microphysics.h -- the C++ plug-in class PAM's drivers instantiate as `Microphysics`
// (selected with -DPAM_MICRO=kessler_amd, physics/micro/CMakeLists.txt), same duck-typed members as the reference's
// physics/micro/kessler/Microphysics.h, forwarding to the C ABI of libpam_amd_awfl.so (include/pam_amd_modules.h).
#pragma once
#include <array>
#include <cmath>
#include <string>

#include "pam_coupler.h"
#include "pam_amd_awfl.h"
#include "pam_amd_modules.h"

class Microphysics {
  double *work = nullptr;     // device scratch: old Exner function (nz*ncol) + the time-step minimum
  size_t work_n = 0;

  static void chk(int rc) { if (rc) endrun(pam_amd_awfl_last_error()); }

  void ensure_work(size_t n) {
    if (n == work_n) return;
    if (work) (void)hipFree(work);
    work = nullptr; work_n = 0;
    if (hipMalloc((void **)&work, n * sizeof(double)) != hipSuccess) endrun("ERROR: kessler scratch allocation failed");
    work_n = n;
  }

 public:
  int static constexpr num_tracers = 3;                                        // Microphysics.h:15
  real R_d = 287., cp_d = 1003., cp_v = 1859., R_v = 461., p0 = 1.e5, grav = 9.81;   // Microphysics.h:66-71
  int static constexpr ID_V = 0, ID_C = 1, ID_R = 2;

  Microphysics() {}
  Microphysics(Microphysics const &) = delete;
  ~Microphysics() { if (work) (void)hipFree(work); }

  static int constexpr get_num_tracers() { return 3; }
  static auto constexpr get_diffused_tracers_indices() { return std::array<int, 3>{ID_V, ID_C, ID_R}; }
  static auto constexpr get_num_diffused_tracers() { return (size_t)3; }

  void init(pam::PamCoupler &coupler) {                                        // Microphysics.h:55-104
    coupler.add_tracer("water_vapor", "Water Vapor", true, true);
    coupler.add_tracer("cloud_liquid", "Cloud liquid", true, true);
    coupler.add_tracer("precip_liquid", "precip_liquid", true, true);
    auto &dm = coupler.get_data_manager_device_readwrite();
    dm.register_and_allocate<real>("precl", "precipitation rate", {coupler.get_ny(), coupler.get_nx(), coupler.get_nens()},
                                   {"y", "x", "nens"});                        // allocations are zero-filled
    coupler.set_option<std::string>("micro", "kessler");
    coupler.set_option<real>("R_d", R_d);
    coupler.set_option<real>("R_v", R_v);
    coupler.set_option<real>("cp_d", cp_d);
    coupler.set_option<real>("cp_v", cp_v);
    coupler.set_option<real>("grav", grav);
    coupler.set_option<real>("p0", p0);
  }

  // rainsplit > 0: sub-cycle count agreed between ensemble shards (see pam_amd_kessler_max_stable_dt)
  void timeStep(pam::PamCoupler &coupler, int rainsplit = 0) {                 // Microphysics.h:120-268
    int nz = coupler.get_nz(), ny = coupler.get_ny(), nx = coupler.get_nx(), nens = coupler.get_nens();
    ensure_work((size_t)nz * ny * nx * nens + 1);
    auto &dm = coupler.get_data_manager_device_readwrite();
    chk(pam_amd_kessler_time_step(nens, nx, ny, nz, dm.get<real, 4>("water_vapor").data(), dm.get<real, 4>("cloud_liquid").data(),
                                  dm.get<real, 4>("precip_liquid").data(), dm.get<real const, 4>("density_dry").data(),
                                  dm.get<real, 4>("temp").data(), dm.get<real, 3>("precl").data(),
                                  dm.get<real const, 2>("vertical_midpoint_height").data(), coupler.get_option<real>("crm_dt"),
                                  R_d, R_v, cp_d, p0, work, nullptr, rainsplit, nullptr));
  }

  real max_stable_dt(pam::PamCoupler &coupler) {
    int nz = coupler.get_nz(), ny = coupler.get_ny(), nx = coupler.get_nx(), nens = coupler.get_nens();
    ensure_work((size_t)nz * ny * nx * nens + 1);
    auto &dm = coupler.get_data_manager_device_readwrite();
    double dt_max;
    chk(pam_amd_kessler_max_stable_dt(nens, nx, ny, nz, dm.get<real const, 4>("precip_liquid").data(),
                                      dm.get<real const, 4>("density_dry").data(),
                                      dm.get<real const, 2>("vertical_midpoint_height").data(), coupler.get_option<real>("crm_dt"),
                                      work, nullptr, &dt_max));
    return dt_max;
  }

  std::string micro_name() const { return "kessler"; }                         // Microphysics.h:467
  void finalize(pam::PamCoupler &coupler) {}
};
