// dynamics/awfl_amd/Dycore.h -- the compile-time plug-in class PAM selects with -DPAM_DYCORE=awfl_amd
// (dynamics/CMakeLists.txt:5-17), forwarding to the C ABI of libpam_amd_awfl.so (include/pam_amd_awfl.h).
//
// It calls only members that exist, with these names and argument lists, in PAM's own pam_core/pam_coupler.h and
// pam_core/DataManager.h (tests/test_boundary_surface.py scans this file against those headers where the reference tree
// is mounted), so the same file compiles inside PAM and against the minimal work-alike in ../../pam_coupler.h (here).
// Members replace, one for one and with the same signatures, those of dynamics/awfl/Dycore.h (line numbers in the
// comments).  The only non-coupler calls are the C ABI and hipMemcpy (host -> DataManager array, where the reference
// uses YAKL's deep_copy_to).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "pam_coupler.h"     // pam::PamCoupler, real, endrun
#include "pam_amd_awfl.h"    // C ABI

class Dycore {
  // the reference's finalize() and compute_time_step() are const members taking the coupler const (awfl/Dycore.h:65,1548);
  // the handle and the pointer scratch are implementation state behind that interface
  mutable pam_amd_awfl_t *h = nullptr;
  mutable std::vector<double *> trc;   // device pointers of the tracer arrays, coupler registration order
  int ncycles_ = 0;

  static void chk(int rc) { if (rc) endrun(pam_amd_awfl_last_error()); }          // pam_const.h:249-252

#ifdef PAM_STANDALONE
  // The reference reads `initData` from the standalone driver's YAML file with yaml-cpp (awfl/Dycore.h:991-1004).  Its input files
  // are flat `key : value  # comment` lists (the .yaml files under standalone/mmf_simplified/inputs), which is all this reads; inside PAM, where
  // yaml-cpp is linked, YAML::LoadFile(inFile)["initData"].as<std::string>() is the same value.
  static std::string yaml_flat_value(std::string const &file, std::string const &key) {
    std::ifstream in(file);
    if (!in) endrun("ERROR: cannot open standalone_input_file " + file);
    std::string line;
    while (std::getline(in, line)) {
      auto hash = line.find('#');
      if (hash != std::string::npos) line.erase(hash);
      auto colon = line.find(':');
      if (colon == std::string::npos) continue;
      auto trim = [](std::string v) {
        auto a = v.find_first_not_of(" \t\r\""), b = v.find_last_not_of(" \t\r\"");
        return a == std::string::npos ? std::string() : v.substr(a, b - a + 1);
      };
      if (trim(line.substr(0, colon)) == key) return trim(line.substr(colon + 1));
    }
    endrun("ERROR: key " + key + " not found in " + file);
    return "";
  }
#endif

  // read-write views of the coupler fields (awfl/Dycore.h:1301-1310): marks the entries dirty like the reference
  pam_amd_awfl_fields_t fields(pam::PamCoupler &coupler) const {
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

  // read-only views (awfl/Dycore.h:74-83,1357-1366): the C ABI struct is shared with the read-write entry points, the
  // entry points these are passed to only read
  pam_amd_awfl_fields_t fields_readonly(pam::PamCoupler const &coupler) const {
    auto &dm = coupler.get_data_manager_device_readonly();                         // pam_coupler.h:64
    trc.clear();
    for (auto &name : coupler.get_tracer_names()) trc.push_back(const_cast<double *>(dm.get<real const, 4>(name).data()));
    pam_amd_awfl_fields_t f;
    f.density_dry = const_cast<double *>(dm.get<real const, 4>("density_dry").data());
    f.uvel = const_cast<double *>(dm.get<real const, 4>("uvel").data());
    f.vvel = const_cast<double *>(dm.get<real const, 4>("vvel").data());
    f.wvel = const_cast<double *>(dm.get<real const, 4>("wvel").data());
    f.temp = const_cast<double *>(dm.get<real const, 4>("temp").data());
    f.tracers = trc.data();
    return f;
  }

 public:
  // awfl/Dycore.h:835
  void init(pam::PamCoupler &coupler, bool verbose = false) {
    auto names = coupler.get_tracer_names();
    int num_tracers = coupler.get_num_tracers();
    std::vector<unsigned char> pos(num_tracers), mass(num_tracers);
    int idWV = -1;
    for (int t = 0; t < num_tracers; t++) {
      std::string desc;
      bool found, p, m;
      coupler.get_tracer_info(names[t], desc, found, p, m);                        // pam_coupler.h:229
      pos[t] = p; mass[t] = m;
      if (names[t] == "water_vapor") idWV = t;                                     // awfl/Dycore.h:969
    }
    auto opt = [&](char const *k) { return coupler.option_exists(k) ? coupler.get_option<real>(k) : (real)NAN; };
    pam_amd_awfl_config_t cfg;
    cfg.nens = coupler.get_nens(); cfg.nx = coupler.get_nx(); cfg.ny = coupler.get_ny(); cfg.nz = coupler.get_nz();
    cfg.num_tracers = num_tracers;
    cfg.xlen = coupler.get_xlen(); cfg.ylen = coupler.get_ylen();
    cfg.R_d = opt("R_d"); cfg.cp_d = opt("cp_d"); cfg.R_v = opt("R_v"); cfg.cp_v = opt("cp_v");
    cfg.p0 = opt("p0"); cfg.grav = opt("grav");                                    // awfl/Dycore.h:871-876
    cfg.cv_d = opt("cv_d"); cfg.gamma_d = opt("gamma_d"); cfg.kappa_d = opt("kappa_d");
    cfg.cv_v = opt("cv_v"); cfg.C0 = opt("C0");                                    // awfl/Dycore.h:883-890
    cfg.idWV = idWV;
    cfg.tracer_positive = pos.data(); cfg.tracer_adds_mass = mass.data();
    cfg.vertical_cell_dz = coupler.get_data_manager_device_readonly().get<real const, 2>("vertical_cell_dz").data();
    cfg.stream = nullptr;                                                          // the default stream of the CURRENT device (what YAKL uses)
    chk(pam_amd_awfl_init(&cfg, &h));
    // what the reference writes back into the coupler (awfl/Dycore.h:866-891,974)
    coupler.set_option<bool>("balance_hydrostasis_with_gravity", true);
    for (char const *k : {"R_d", "cp_d", "R_v", "cp_v", "p0", "grav", "cv_d", "gamma_d", "kappa_d", "cv_v", "C0"}) {
      double v;
      chk(pam_amd_awfl_get_option(h, k, &v));
      if (!coupler.option_exists(k)) coupler.set_option<real>(k, v);
    }
    coupler.set_option<int>("idWV", idWV);
    // the dycore's DataManager entries: allocated and owned by the DataManager exactly as in the reference
    // (awfl/Dycore.h:868,897-898,983-984), so they outlive finalize(); the kernels are bound to that storage
    auto &dm = coupler.get_data_manager_device_readwrite();
    for (char const *name : {"variable_gravity", "hy_dens_cells", "hy_pressure_cells", "vert_sten_to_coefs",
                             "vert_weno_recon_lower"}) {
      double *p;
      int dims[5], nd;
      chk(pam_amd_awfl_get_array(h, name, &p, dims, &nd));
      dm.register_and_allocate<real>(name, "", std::vector<int>(dims, dims + nd));   // DataManager.h:91
      if (nd == 2) chk(pam_amd_awfl_bind_array(h, name, dm.get<real, 2>(name).data()));
      if (nd == 4) chk(pam_amd_awfl_bind_array(h, name, dm.get<real, 4>(name).data()));
      if (nd == 5) chk(pam_amd_awfl_bind_array(h, name, dm.get<real, 5>(name).data()));
    }
    // awfl/Dycore.h:975-981 (bool entries; the reference fills them with deep_copy_to)
    std::vector<char> pos_b(num_tracers), mass_b(num_tracers);
    static_assert(sizeof(bool) == 1, "bool entries are copied bytewise");
    for (int t = 0; t < num_tracers; t++) { pos_b[t] = pos[t] ? 1 : 0; mass_b[t] = mass[t] ? 1 : 0; }
    dm.register_and_allocate<bool>("tracer_adds_mass", "", {num_tracers});
    if (hipMemcpy(dm.get<bool, 1>("tracer_adds_mass").data(), mass_b.data(), num_tracers, hipMemcpyHostToDevice) != hipSuccess)
      endrun("ERROR: copying tracer_adds_mass to the device failed");
    dm.register_and_allocate<bool>("tracer_positive", "", {num_tracers});
    if (hipMemcpy(dm.get<bool, 1>("tracer_positive").data(), pos_b.data(), num_tracers, hipMemcpyHostToDevice) != hipSuccess)
      endrun("ERROR: copying tracer_positive to the device failed");
#ifdef PAM_DEBUG
    chk(pam_amd_awfl_set_debug_conservation(h, 1));                                // awfl/Dycore.h:136-138,224-251
#endif
    // idealised initial data of the standalone driver (awfl/Dycore.h:986-1090: "thermal" :1021-1088, "supercell" :1096-1276),
    // filled on the device; "external" (the MMF case, and what the absence of the option means): nothing to do (:1005-1011)
    if (coupler.option_exists("standalone_input_file")) {
#ifdef PAM_STANDALONE
      std::string inFile = coupler.get_option<std::string>("standalone_input_file");
      std::string dataStr = yaml_flat_value(inFile, "initData");
      if (dataStr != "thermal" && dataStr != "supercell" && dataStr != "external") endrun("ERROR: Invalid data_spec");   // :1002
      auto f = fields(coupler);
      chk(pam_amd_awfl_init_idealized(h, &f, dataStr.c_str(), dm.get<real const, 2>("vertical_midpoint_height").data(),
                                      dm.get<real const, 2>("vertical_interface_height").data()));
#endif
    }
    (void)verbose;
  }

  // awfl/Dycore.h:107
  void timeStep(pam::PamCoupler &coupler) {
    chk(sync_balance_option(coupler));
    auto f = fields(coupler);
    chk(pam_amd_awfl_time_step(h, &f, coupler.get_option<real>("crm_dt"), /*dt_dyn_hint=*/0., &ncycles_, nullptr));
    report_conservation();
  }

  // Not in the reference (one process, one device): the ensemble sharded by member index over several devices, one coupler and
  // one Dycore per device.  The dynamics step is a minimum over ALL members (awfl/Dycore.h:86-101,141-145): the host takes the
  // minimum of the devices' compute_time_step() values (N host doubles; examples/driver.cpp --gpus N) and passes it in, so every
  // shard sub-cycles as the unsharded ensemble would.
  void timeStep(pam::PamCoupler &coupler, real dt_dyn_all_members) {
    chk(sync_balance_option(coupler));
    auto f = fields(coupler);
    chk(pam_amd_awfl_time_step(h, &f, coupler.get_option<real>("crm_dt"), dt_dyn_all_members, &ncycles_, nullptr));
    report_conservation();
  }
  int last_ncycles() const { return ncycles_; }        // sub-cycles of the most recent timeStep (awfl/Dycore.h:144)
  // The reference's PAM_DEBUG self-check (awfl/Dycore.h:36-58,136-138,224-251) at run time: the mass of every variable and member
  // before / after each timeStep; violations (> 1e-10 relative and absolute) are printed as the reference prints them
  void set_debug_conservation(bool on) { chk(pam_amd_awfl_set_debug_conservation(h, on ? 1 : 0)); debug_mass_ = on; }
  int conservation_violations() const {
    int n = 0;
    if (debug_mass_) chk(pam_amd_awfl_get_conservation(h, &n, nullptr, nullptr, nullptr));
    return n;
  }
  real conservation_max_rel_diff() const {
    double r = 0;
    if (debug_mass_) chk(pam_amd_awfl_get_conservation(h, nullptr, &r, nullptr, nullptr));
    return r;
  }

  // awfl/Dycore.h:65
  real compute_time_step(pam::PamCoupler const &coupler, real cfl = 0.8) const {
    auto f = fields_readonly(coupler);
    double dt;
    chk(pam_amd_awfl_compute_time_step(h, &f, cfl, &dt));
    return dt;
  }

  // awfl/Dycore.h:1392
  void declare_current_profile_as_hydrostatic(pam::PamCoupler &coupler, bool use_gcm_data = false) const {
    chk(sync_balance_option(coupler));
    auto f = fields_readonly(coupler);
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

  // awfl/Dycore.h:1336-1338 / :1281-1283, the reference's argument lists: the caller's halo'd arrays
  // state(5, nz+2hs, ny+2hs, nx+2hs, nens) and tracers(num_tracers, ...), hs = 3 (what E3SM's pam_driver passes)
  void convert_coupler_to_dynamics(pam::PamCoupler &coupler, real5d &state, real5d &tracers) const {
    auto f = fields_readonly(coupler);
    chk(pam_amd_awfl_convert_coupler_to_dynamics_arrays(h, &f, state.data(), tracers.data()));
  }
  void convert_dynamics_to_coupler(pam::PamCoupler &coupler, realConst5d state, realConst5d tracers) const {
    auto f = fields(coupler);
    chk(pam_amd_awfl_convert_dynamics_to_coupler_arrays(h, &f, state.data(), tracers.data()));
  }
  // ... and without them: the dynamics state resident inside the handle is refreshed from / written to
  // the coupler
  void convert_coupler_to_dynamics(pam::PamCoupler &coupler) const {
    auto f = fields_readonly(coupler);
    chk(pam_amd_awfl_convert_coupler_to_dynamics(h, &f));
  }
  void convert_dynamics_to_coupler(pam::PamCoupler &coupler) const {
    auto f = fields(coupler);
    chk(pam_amd_awfl_convert_dynamics_to_coupler(h, &f));
  }

  char const *dycore_name() const { return pam_amd_awfl_dycore_name(h); }           // awfl/Dycore.h:1544

  // awfl/Dycore.h:1548 (empty there).  The DataManager keeps the dycore's entries -- it owns them -- and frees them in
  // its own finalize (DataManager.h:596-602); only the handle (scratch arrays, streams) is released here.
  void finalize(pam::PamCoupler const &coupler) const {
    if (!h) return;
    pam_amd_awfl_finalize(h);
    h = nullptr;
  }

 private:
#ifdef PAM_DEBUG
  bool debug_mass_ = true;
#else
  bool debug_mass_ = false;
#endif
  void report_conservation() const {
    if (!debug_mass_) return;
    int n = 0;
    chk(pam_amd_awfl_get_conservation(h, &n, nullptr, nullptr, nullptr));
    if (n > 0) std::cout << pam_amd_awfl_conservation_report(h);                  // awfl/Dycore.h:240-247
  }
  // the reference re-reads the option on every call (awfl/Dycore.h:284,624,1410)
  int sync_balance_option(pam::PamCoupler const &coupler) const {
    double cur;
    int rc = pam_amd_awfl_get_option(h, "balance_hydrostasis_with_gravity", &cur);
    if (rc) return rc;
    bool want = coupler.get_option<bool>("balance_hydrostasis_with_gravity");
    if (want != (cur != 0)) return pam_amd_awfl_set_balance_hydrostasis_with_gravity(h, want ? 1 : 0);
    return 0;
  }
};
