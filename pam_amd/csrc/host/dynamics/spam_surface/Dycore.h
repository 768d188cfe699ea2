// dynamics/spam_surface/Dycore.h -- BASELINE config C5 ("SPAM dycore swap-in behind the same pam_coupler plug-in
// surface"), boundary only: a second `class Dycore` exposing the member set of the reference's SPAM dycore
// (dynamics/spam/Dycore.h: init :79, pre_time_loop :169, update_dt :233, compute_time_step :244, timeStep :248,
// finalize :325, dycore_name :327), selected exactly like any other dycore -- by the include path
// (dynamics/CMakeLists.txt:5-17) -- and driven by the same examples/driver.cpp with -DPAMC_DYCORE as the reference
// driver is (standalone/mmf_simplified/driver.cpp:225-227).
//
// SPAM's numerics (Hamiltonian / discrete-exterior-calculus discretisation, dynamics/spam/src) are OUT OF SCOPE
// (DESIGN.md section 7): this class advances nothing.  What it proves is that the coupler surface of
// this repository (pam_coupler.h work-alike, run_module, options, DataManager) is dycore-agnostic: the driver, the
// modules and the microphysics compile and run unchanged against a Dycore with SPAM's extra members and SPAM's
// signatures (finalize takes the coupler non-const there; compute_time_step returns 0 and is non-const).
#pragma once
#include <string>

#include "pam_coupler.h"

class Dycore {
  real dtcrm = 0;          // spam/Dycore.h:239-240 params.dt_crm_phys / params.dtcrm
  int crm_per_phys = 1;
  int prevstep = 0;        // spam/Dycore.h:230
  long steps = 0;

 public:
  void init(pam::PamCoupler &coupler, bool verbose = false) {                        // spam/Dycore.h:79
    // SPAM reads its parameters from the coupler options and the grid getters (spam/Dycore.h:96-130)
    dtcrm = coupler.get_option<real>("crm_dt");
    if (coupler.get_nens() < 1 || coupler.get_nz() < 1) endrun("ERROR: coupler state not allocated before dycore.init");
    if (!coupler.option_exists("spam_crm_per_phys")) coupler.set_option<int>("spam_crm_per_phys", 1);
    crm_per_phys = coupler.get_option<int>("spam_crm_per_phys");
  }

  void pre_time_loop(pam::PamCoupler &coupler) { prevstep = 1; }                      // spam/Dycore.h:169,230

  void update_dt(pam::PamCoupler &coupler) {                                         // spam/Dycore.h:233-241
    dtcrm = coupler.get_option<real>("crm_dt") / crm_per_phys;
  }

  real compute_time_step(pam::PamCoupler const &coupler, real cfl_in = -1) { return 0.; }   // spam/Dycore.h:244-246

  void timeStep(pam::PamCoupler &coupler) {                                          // spam/Dycore.h:248
    if (prevstep != 1) endrun("ERROR: dycore.pre_time_loop(coupler) must be called before timeStep (PAMC_DYCORE)");
    // touches the coupler fields the way a dycore does (read-write access marks them dirty for run_module's tracing)
    auto &dm = coupler.get_data_manager_device_readwrite();
    (void)dm.get<real, 4>("density_dry");
    (void)dm.get<real, 4>("uvel");
    (void)dm.get<real, 4>("wvel");
    (void)dm.get<real, 4>("temp");
    steps++;
  }

  void finalize(pam::PamCoupler &coupler) {}                                         // spam/Dycore.h:325

  const char *dycore_name() const { return "SPAM++ (surface stub: numerics out of scope)"; }   // spam/Dycore.h:327
  long steps_taken() const { return steps; }
};
