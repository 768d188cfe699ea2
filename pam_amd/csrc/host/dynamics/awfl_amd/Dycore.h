// dynamics/awfl_amd/Dycore.h -- the compile-time plug-in class PAM selects with -DPAM_DYCORE=awfl_amd
// (dynamics/CMakeLists.txt:5-17), forwarding to the C ABI of libpam_amd_awfl.so (include/pam_amd_awfl.h).
//
// It uses only members that exist with the same names in PAM's own pam_core/pam_coupler.h / DataManager.h, so the same
// file compiles against the real PAM headers (inside PAM) and against the minimal work-alike in ../../pam_coupler.h (here).
// Members replace, one for one, those of dynamics/awfl/Dycore.h (line numbers in the comments).
#pragma once
#include <cmath>
#include <string>
#include <vector>

#include "pam_coupler.h"     // pam::PamCoupler, real, endrun
#include "pam_amd_awfl.h"    // C ABI

class Dycore {
  pam_amd_awfl_t *h = nullptr;
  std::vector<double *> trc;   // device pointers of the tracer arrays, coupler registration order

  static void chk(int rc) { if (rc) endrun(pam_amd_awfl_last_error()); }          // pam_const.h:249-252

  pam_amd_awfl_fields_t fields(pam::PamCoupler &coupler) {
    auto &dm = coupler.get_data_manager_device_readwrite();                        // pam_coupler.h:65
    trc.clear();
    for (auto &name : coupler.get_tracer_names()) trc.push_back(dm.get<real, 4>(name).data());
    pam_amd_awfl_fields_t f;
    f.density_dry = dm.get<real, 4>("density_dry").data();
    f.uvel = dm.get<real, 4>("uvel").data();
    f.vvel = dm.get<real, 4>("vvel").data();
    f.wvel = dm.get<real, 4>("wvel").data();
    f.temp = dm.get<real, 4>("temp").data();
    f.tracers = trc.data();
    return f;
  }

 public:
  // awfl/Dycore.h:835
  void init(pam::PamCoupler &coupler, bool verbose = false) {
    auto names = coupler.get_tracer_names();
    std::vector<unsigned char> pos(names.size()), mass(names.size());
    int idWV = -1;
    for (size_t t = 0; t < names.size(); t++) {
      std::string desc;
      bool found, p, m;
      coupler.get_tracer_info(names[t], desc, found, p, m);                        // pam_coupler.h:229
      pos[t] = p; mass[t] = m;
      if (names[t] == "water_vapor") idWV = (int)t;                                // awfl/Dycore.h:969
    }
    auto opt = [&](char const *k) { return coupler.option_exists(k) ? coupler.get_option<real>(k) : (real)NAN; };
    pam_amd_awfl_config_t cfg;
    cfg.nens = coupler.get_nens(); cfg.nx = coupler.get_nx(); cfg.ny = coupler.get_ny(); cfg.nz = coupler.get_nz();
    cfg.num_tracers = (int)names.size();
    cfg.xlen = coupler.get_xlen(); cfg.ylen = coupler.get_ylen();
    cfg.R_d = opt("R_d"); cfg.cp_d = opt("cp_d"); cfg.R_v = opt("R_v"); cfg.cp_v = opt("cp_v");
    cfg.p0 = opt("p0"); cfg.grav = opt("grav");
    cfg.idWV = idWV;
    cfg.tracer_positive = pos.data(); cfg.tracer_adds_mass = mass.data();
    cfg.vertical_cell_dz = coupler.get_data_manager_device_readonly().get<real const, 2>("vertical_cell_dz").data();
    cfg.stream = nullptr;                                                          // the default stream (what YAKL uses)
    chk(pam_amd_awfl_init(&cfg, &h));
    // what the reference writes back into the coupler (awfl/Dycore.h:866-891,974-984)
    coupler.set_option<bool>("balance_hydrostasis_with_gravity", true);
    for (char const *k : {"R_d", "cp_d", "R_v", "cp_v", "p0", "grav", "cv_d", "gamma_d", "kappa_d", "cv_v", "C0"}) {
      double v;
      chk(pam_amd_awfl_get_option(h, k, &v));
      if (!coupler.option_exists(k)) coupler.set_option<real>(k, v);
    }
    coupler.set_option<int>("idWV", idWV);
    auto &dm = coupler.get_data_manager_device_readwrite();
    for (char const *name : {"variable_gravity", "hy_dens_cells", "hy_pressure_cells", "vert_sten_to_coefs",
                             "vert_weno_recon_lower"}) {
      double *p;
      int dims[5], nd;
      chk(pam_amd_awfl_get_array(h, name, &p, dims, &nd));
      dm.register_existing<real>(name, "", std::vector<int>(dims, dims + nd), p);   // DataManager.h:158
    }
  }

  // awfl/Dycore.h:107
  void timeStep(pam::PamCoupler &coupler) {
    chk(pam_amd_awfl_set_balance_hydrostasis_with_gravity_if_changed(coupler));
    auto f = fields(coupler);
    chk(pam_amd_awfl_time_step(h, &f, coupler.get_option<real>("crm_dt"), /*dt_dyn_hint=*/0., nullptr, nullptr));
  }

  // awfl/Dycore.h:65
  real compute_time_step(pam::PamCoupler &coupler, real cfl = 0.8) {
    auto f = fields(coupler);
    double dt;
    chk(pam_amd_awfl_compute_time_step(h, &f, cfl, &dt));
    return dt;
  }

  // awfl/Dycore.h:1392
  void declare_current_profile_as_hydrostatic(pam::PamCoupler &coupler, bool use_gcm_data = false) {
    chk(pam_amd_awfl_set_balance_hydrostasis_with_gravity_if_changed(coupler));
    auto f = fields(coupler);
    if (!use_gcm_data) { chk(pam_amd_awfl_declare_current_profile_as_hydrostatic(h, &f, nullptr)); return; }
    auto &dm = coupler.get_data_manager_device_readonly();
    pam_amd_awfl_gcm_columns_t g;
    g.gcm_density_dry = dm.get<real const, 2>("gcm_density_dry").data();
    g.gcm_temp = dm.get<real const, 2>("gcm_temp").data();
    g.gcm_water_vapor = dm.get<real const, 2>("gcm_water_vapor").data();
    g.gcm_cloud_water = dm.get<real const, 2>("gcm_cloud_water").data();
    g.gcm_cloud_ice = dm.get<real const, 2>("gcm_cloud_ice").data();
    chk(pam_amd_awfl_declare_current_profile_as_hydrostatic(h, &f, &g));
  }

  // awfl/Dycore.h:1336 / :1281
  void convert_coupler_to_dynamics(pam::PamCoupler &c) { auto f = fields(c); chk(pam_amd_awfl_convert_coupler_to_dynamics(h, &f)); }
  void convert_dynamics_to_coupler(pam::PamCoupler &c) { auto f = fields(c); chk(pam_amd_awfl_convert_dynamics_to_coupler(h, &f)); }

  char const *dycore_name() const { return pam_amd_awfl_dycore_name(h); }           // awfl/Dycore.h:1544

  void finalize(pam::PamCoupler &coupler) {                                         // awfl/Dycore.h:1548
    if (!h) return;
    auto &dm = coupler.get_data_manager_device_readwrite();
    for (char const *name : {"variable_gravity", "hy_dens_cells", "hy_pressure_cells", "vert_sten_to_coefs",
                             "vert_weno_recon_lower"})
      dm.unregister(name);
    pam_amd_awfl_finalize(h);
    h = nullptr;
  }

 private:
  // the reference re-reads the option on every call (awfl/Dycore.h:284,624,1410)
  int pam_amd_awfl_set_balance_hydrostasis_with_gravity_if_changed(pam::PamCoupler &coupler) {
    double cur;
    int rc = pam_amd_awfl_get_option(h, "balance_hydrostasis_with_gravity", &cur);
    if (rc) return rc;
    bool want = coupler.get_option<bool>("balance_hydrostasis_with_gravity");
    if (want != (cur != 0)) return pam_amd_awfl_set_balance_hydrostasis_with_gravity(h, want ? 1 : 0);
    return 0;
  }
};
