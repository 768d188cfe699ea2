// examples/driver.cpp -- a small C++ host driver over the plug-in surface, mirroring the call sequence of the
// reference's standalone/mmf_simplified/driver.cpp (:120-191 set-up, :237-272 time loop) for the dycore alone:
//
//   coupler.allocate_coupler_state -> set_grid -> [micro.init: tracer registration + constants] -> dycore.init
//   -> (host model fills the coupler fields) -> dycore.declare_current_profile_as_hydrostatic
//   -> N x { coupler.run_module("dycore", dycore.timeStep); [sponge_layer]; [micro.timeStep] } -> output
//
// Input/output are raw little-endian fp64 files written/read by tests/test_cpp_driver.py (the reference reads YAML and
// writes netCDF; neither library exists in this image and I/O is out of scope):
//   header (8 x int64): nens nx ny nz num_tracers nsteps flags has_consts ; then xlen ylen crm_dt (3 x f64),
//   (flags: bit0 = hydrostasis mode A, bit1 = run modules::sponge_layer, bit2 = Kessler Microphysics: its init registers
//   the three water tracers, so num_tracers must be 3, and "precl" (ny*nx*nens) is appended to the output)
//   6 constants (R_d cp_d R_v cp_v p0 grav), zint (nz+1), tracer flags (num_tracers x 2 bytes positive/adds_mass,
//   then idWV int64), then density_dry,uvel,vvel,wvel,temp,(tracers...) each nz*ny*nx*nens f64.
#include <cstdint>
#include <cstdio>
#include <fstream>

// The dycore is selected by the include path, as in PAM (dynamics/CMakeLists.txt:5-17: -DPAM_DYCORE=<dir> puts
// dynamics/<dir> first): -Ipam_amd/csrc/host/dynamics/awfl_amd for the MI355X AWFL step, .../dynamics/spam_surface for
// the SPAM-surface stub of BASELINE config C5 (built with -DPAMC_DYCORE, like the reference's SPAM builds).
#include "Dycore.h"
#include "modules/gcm_forcing.h"     // compiled here; exercised from Python (tests/test_modules.py)
#include "modules/sponge_layer.h"
#include "physics/micro/kessler_amd/Microphysics.h"

static void die(const char *m) { std::fprintf(stderr, "driver: %s\n", m); std::exit(2); }

int main(int argc, char **argv) {
  if (argc != 3) die("usage: driver <input.bin> <output.bin>");
  std::ifstream in(argv[1], std::ios::binary);
  if (!in) die("cannot open input");
  int64_t hdr[8];
  in.read((char *)hdr, sizeof(hdr));
  const int nens = hdr[0], nx = hdr[1], ny = hdr[2], nz = hdr[3], nt = hdr[4], nsteps = hdr[5];
  const bool mode_a = (hdr[6] & 1) != 0, with_sponge = (hdr[6] & 2) != 0, with_micro = (hdr[6] & 4) != 0;
  if (with_micro && nt != 3) die("the Kessler microphysics registers exactly 3 tracers");
  double geo[3], consts[6];
  in.read((char *)geo, sizeof(geo));
  in.read((char *)consts, sizeof(consts));
  std::vector<real> zint(nz + 1);
  in.read((char *)zint.data(), zint.size() * sizeof(real));
  std::vector<unsigned char> flags(2 * nt);
  in.read((char *)flags.data(), flags.size());
  int64_t idWV;
  in.read((char *)&idWV, sizeof(idWV));
  const size_t ncell = (size_t)nz * ny * nx * nens;
  try {
    pam::PamCoupler coupler;
    coupler.set_option<real>("crm_dt", geo[2]);
    coupler.allocate_coupler_state(nz, ny, nx, nens);                       // driver.cpp:177
    coupler.set_grid(geo[0], geo[1], zint);                                 // driver.cpp:180
    // what micro.init()/sgs.init() do for the dycore: constants + tracer registration, BEFORE dycore.init (driver.cpp:189-191)
    const char *cn[6] = {"R_d", "cp_d", "R_v", "cp_v", "p0", "grav"};
    if (hdr[7]) for (int i = 0; i < 6; i++) coupler.set_option<real>(cn[i], consts[i]);
    Microphysics micro;
    if (with_micro) {
      micro.init(coupler);                                                   // driver.cpp:189
    } else {
      for (int t = 0; t < nt; t++)
        coupler.add_tracer(t == idWV ? "water_vapor" : "tracer_" + std::to_string(t), "", flags[2 * t] != 0, flags[2 * t + 1] != 0);
    }
    Dycore dycore;
    dycore.init(coupler);                                                    // driver.cpp:191
    std::printf("Dycore: %s\n", dycore.dycore_name());                       // driver.cpp:203
    auto &dm = coupler.get_data_manager_device_readwrite();
    std::vector<real> buf(ncell);
    std::vector<std::string> names = {"density_dry", "uvel", "vvel", "wvel", "temp"};
    for (auto &n : coupler.get_tracer_names()) names.push_back(n);
    for (auto &n : names) {
      in.read((char *)buf.data(), ncell * sizeof(real));
      if (!in) die("short input file");
      if (hipMemcpy(dm.get<real, 4>(n).data(), buf.data(), ncell * sizeof(real), hipMemcpyHostToDevice) != hipSuccess) die("memcpy");
    }
#ifdef PAMC_DYCORE
    (void)mode_a;
    dycore.pre_time_loop(coupler);                                           // driver.cpp:225-227
#else
    if (!mode_a) coupler.set_option<bool>("balance_hydrostasis_with_gravity", false);   // after init(), SURVEY 8c
    dycore.declare_current_profile_as_hydrostatic(coupler);                  // the host model does this once per GCM step
    if (hdr[6] & 8) {
      // the two converts with the reference's own argument lists (awfl/Dycore.h:1336-1338, :1281-1283), as E3SM's pam_driver
      // calls them: coupler -> the caller's halo'd arrays, coupler fields wiped, arrays -> coupler
      const int hs = 3;
      const std::vector<int> hdims = {nz + 2 * hs, ny + 2 * hs, nx + 2 * hs, nens};
      size_t nh = 1;
      for (int d : hdims) nh *= d;
      real *ps = nullptr, *pt = nullptr;
      if (hipMalloc((void **)&ps, 5 * nh * sizeof(real)) != hipSuccess || hipMalloc((void **)&pt, (size_t)nt * nh * sizeof(real)) != hipSuccess) die("hipMalloc");
      real5d state(ps, {5, hdims[0], hdims[1], hdims[2], hdims[3]}), tracers(pt, {nt, hdims[0], hdims[1], hdims[2], hdims[3]});
      dycore.convert_coupler_to_dynamics(coupler, state, tracers);
      for (auto &n : names) (void)hipMemsetAsync(dm.get<real, 4>(n).data(), 0xFF, ncell * sizeof(real), 0);   // NaN bit patterns
      dycore.convert_dynamics_to_coupler(coupler, realConst5d(ps, state.dims()), realConst5d(pt, tracers.dims()));
      if (hipDeviceSynchronize() != hipSuccess) die("device error");
      (void)hipFree(ps); (void)hipFree(pt);
    }
#endif
    for (int s = 0; s < nsteps; s++) {
      coupler.run_module("dycore", [&](pam::PamCoupler &c) { dycore.timeStep(c); });    // driver.cpp:248
      if (with_sponge) coupler.run_module("sponge_layer", modules::sponge_layer);       // driver.cpp:250
      if (with_micro) coupler.run_module("micro", [&](pam::PamCoupler &c) { micro.timeStep(c); });   // driver.cpp:253
    }
    if (hipDeviceSynchronize() != hipSuccess) die("device error");
    std::ofstream out(argv[2], std::ios::binary);
    for (auto &n : names) {
      if (hipMemcpy(buf.data(), dm.get<real, 4>(n).data(), ncell * sizeof(real), hipMemcpyDeviceToHost) != hipSuccess) die("memcpy");
      out.write((char *)buf.data(), ncell * sizeof(real));
    }
    if (with_micro) {
      const size_t n2 = (size_t)ny * nx * nens;
      if (hipMemcpy(buf.data(), dm.get<real, 3>("precl").data(), n2 * sizeof(real), hipMemcpyDeviceToHost) != hipSuccess) die("memcpy");
      out.write((char *)buf.data(), n2 * sizeof(real));
    }
    dycore.finalize(coupler);                                                // driver.cpp:285
  } catch (std::string &msg) {
    std::fprintf(stderr, "driver: endrun: %s\n", msg.c_str());
    return 1;
  }
  return 0;
}
