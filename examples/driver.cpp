// examples/driver.cpp -- a small C++ host driver over the plug-in surface, mirroring the call sequence of the
// reference's standalone/mmf_simplified/driver.cpp (:120-191 set-up, :237-272 time loop) for the dycore alone:
//
//   coupler.allocate_coupler_state -> set_grid -> [micro.init: tracer registration + constants] -> dycore.init
//   -> (host model fills the coupler fields) -> dycore.declare_current_profile_as_hydrostatic
//   -> N x { coupler.run_module("dycore", dycore.timeStep); [sponge_layer]; [micro.timeStep] } -> output
//
//   driver [--gpus N] [--tile R] [--bench K W] <input.bin> <output.bin>
//
// --gpus N: the ensemble is sharded by member index over N devices of this node -- ONE host thread, ONE coupler and ONE dycore
// handle per device (hipSetDevice before init), no inter-device halo and no collective library: the exchanges the reference
// semantics need are the two sub-cycling steps, each a minimum over ALL members -- the dynamics time step (awfl/Dycore.h:86-101,
// 141-145) and, with the Kessler microphysics, its sedimentation step (kessler/Microphysics.h:385-390) -- taken here over N host
// doubles behind a barrier (HostMin) and handed to Dycore::timeStep(coupler, dt_dyn) / Microphysics::timeStep(coupler, rainsplit).  With fewer devices than ranks the ranks share
// devices (rehearsal on a 1-GPU box; bit-identical results, tests/test_cpp_driver.py).
// --tile R: the input's members are repeated R times along nens (tile t gets +t mK on temp so that no two CRMs are equal).
// --bench K W: W untimed + K timed steps between barriers; rank 0 prints one JSON line with the wall time (bench.py --launcher cpp).
//
// Input/output are raw little-endian fp64 files written/read by tests/test_cpp_driver.py (the reference reads YAML and
// writes netCDF; neither library exists in this image and I/O is out of scope):
//   header (8 x int64): nens nx ny nz num_tracers nsteps flags has_consts ; then xlen ylen crm_dt (3 x f64),
//   (flags: bit0 = hydrostasis mode A, bit1 = run modules::sponge_layer, bit2 = Kessler Microphysics: its init registers
//   the three water tracers, so num_tracers must be 3, and "precl" (ny*nx*nens) is appended to the output)
//   6 constants (R_d cp_d R_v cp_v p0 grav), zint (nz+1), tracer flags (num_tracers x 2 bytes positive/adds_mass,
//   then idWV int64), then density_dry,uvel,vvel,wvel,temp,(tracers...) each nz*ny*nx*nens f64.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <mutex>
#include <thread>

// The dycore is selected by the include path, as in PAM (dynamics/CMakeLists.txt:5-17: -DPAM_DYCORE=<dir> puts
// dynamics/<dir> first): -Ipam_amd/csrc/host/dynamics/awfl_amd for the MI355X AWFL step, .../dynamics/spam_surface for
// the SPAM-surface stub of BASELINE config C5 (built with -DPAMC_DYCORE, like the reference's SPAM builds).
#include "Dycore.h"
#include "modules/gcm_forcing.h"     // compiled here; exercised from Python (tests/test_modules.py)
#include "modules/sponge_layer.h"
#include "modules/broadcast_initial_gcm_column.h"
#include "modules/perturb_temperature.h"
#include "physics/micro/kessler_amd/Microphysics.h"

#include <map>
#include <sstream>

static void die(const char *m) { std::fprintf(stderr, "driver: %s\n", m); std::exit(2); }

// min over the ranks' values: N host doubles behind a barrier.  A rank that fails releases the others (they throw too).
class HostMin {
  std::mutex m;
  std::condition_variable cv;
  const int n;
  int arrived = 0;
  long gen = 0;
  bool failed = false;
  std::vector<double> v;
  double result = 0;
 public:
  explicit HostMin(int n_) : n(n_), v(n_, 0.0) {}
  double operator()(int rank, double x) {
    std::unique_lock<std::mutex> lk(m);
    if (failed) throw std::string("another rank failed");
    v[rank] = x;
    const long g = gen;
    if (++arrived == n) {
      result = *std::min_element(v.begin(), v.end());
      arrived = 0;
      gen++;
      cv.notify_all();
      return result;
    }
    cv.wait(lk, [&] { return gen != g || failed; });
    if (failed) throw std::string("another rank failed");
    return result;
  }
  void barrier(int rank) { (void)(*this)(rank, 0.0); }
  void fail() {
    std::lock_guard<std::mutex> lk(m);
    failed = true;
    cv.notify_all();
  }
};

// members [lo, hi) of rank r (blocks differ by at most one member; pam_amd/parallel.py: shard_range)
static void shard_range(int nens, int r, int n, int &lo, int &hi) {
  const int base = nens / n, rem = nens % n;
  lo = r * base + std::min(r, rem);
  hi = lo + base + (r < rem ? 1 : 0);
}

struct Job {
  int nens, nx, ny, nz, nt, nsteps;
  bool mode_a, with_sponge, with_micro, halo_roundtrip, has_consts;
  double geo[3], consts[6];
  std::vector<real> zint;
  std::vector<unsigned char> flags;
  int64_t idWV;
  int nens_in = 0;                           // members in the input file; member e of the run is input member e % nens_in, tile e / nens_in
  std::vector<std::vector<real>> raw;        // the input: density_dry, uvel, vvel, wvel, temp, tracers...: (nz,ny,nx,nens_in) each
  bool want_output = true;
  std::vector<std::vector<real>> fields;     // the output, same order: (nz,ny,nx,nens) each (only when want_output)
  std::vector<real> precl;                   // (ny,nx,nens), Kessler only
  int bench_steps = 0, bench_warmup = 0;
};

struct BenchResult { double seconds = 0; long substeps = 0; };

// One rank = one device, one coupler, one dycore: the reference driver's call sequence on the members [lo, hi).
static void run_rank(Job &J, int rank, int world, int ndev, HostMin &hmin, BenchResult &bench, std::string &name_out) {
  if (hipSetDevice(rank % ndev) != hipSuccess) endrun("hipSetDevice failed");
  int lo, hi;
  shard_range(J.nens, rank, world, lo, hi);
  const int ne = hi - lo, nx = J.nx, ny = J.ny, nz = J.nz, nt = J.nt;
  if (ne < 1) endrun("ERROR: more ranks than ensemble members");
  const size_t ncol = (size_t)nz * ny * nx, ncell = ncol * ne;
  pam::PamCoupler coupler;
  coupler.set_option<real>("crm_dt", J.geo[2]);
  coupler.allocate_coupler_state(nz, ny, nx, ne);                          // driver.cpp:177
  coupler.set_grid(J.geo[0], J.geo[1], J.zint);                            // driver.cpp:180
  // what micro.init()/sgs.init() do for the dycore: constants + tracer registration, BEFORE dycore.init (driver.cpp:189-191)
  const char *cn[6] = {"R_d", "cp_d", "R_v", "cp_v", "p0", "grav"};
  if (J.has_consts) for (int i = 0; i < 6; i++) coupler.set_option<real>(cn[i], J.consts[i]);
  Microphysics micro;
  if (J.with_micro) {
    micro.init(coupler);                                                   // driver.cpp:189
  } else {
    for (int t = 0; t < nt; t++)
      coupler.add_tracer(t == J.idWV ? "water_vapor" : "tracer_" + std::to_string(t), "", J.flags[2 * t] != 0, J.flags[2 * t + 1] != 0);
  }
  Dycore dycore;
  dycore.init(coupler);                                                    // driver.cpp:191
  if (rank == 0) name_out = dycore.dycore_name();                          // driver.cpp:203
  auto &dm = coupler.get_data_manager_device_readwrite();
  std::vector<real> buf(ncell);
  std::vector<std::string> names = {"density_dry", "uvel", "vvel", "wvel", "temp"};
  for (auto &n : coupler.get_tracer_names()) names.push_back(n);
  for (size_t f = 0; f < names.size(); f++) {                              // this rank's members of every field
    const real *src = J.raw[f].data();
    for (size_t c = 0; c < ncol; c++)
      for (int e = 0; e < ne; e++) {
        const int eg = lo + e, tile_i = eg / J.nens_in;                    // tile t gets +t mK on temp: no two CRMs are equal
        buf[c * ne + e] = src[c * J.nens_in + (eg - tile_i * J.nens_in)] + ((f == 4) ? 1.0e-3 * tile_i : 0.0);
      }
    if (hipMemcpy(dm.get<real, 4>(names[f]).data(), buf.data(), ncell * sizeof(real), hipMemcpyHostToDevice) != hipSuccess) endrun("memcpy");
  }
#ifdef PAMC_DYCORE
  dycore.pre_time_loop(coupler);                                           // driver.cpp:225-227
#else
  if (!J.mode_a) coupler.set_option<bool>("balance_hydrostasis_with_gravity", false);   // after init(), SURVEY 8c
  dycore.declare_current_profile_as_hydrostatic(coupler);                  // the host model does this once per GCM step
  if (J.halo_roundtrip) {
    // the two converts with the reference's own argument lists (awfl/Dycore.h:1336-1338, :1281-1283), as E3SM's pam_driver
    // calls them: coupler -> the caller's halo'd arrays, coupler fields wiped, arrays -> coupler
    const int hs = 3;
    const std::vector<int> hdims = {nz + 2 * hs, ny + 2 * hs, nx + 2 * hs, ne};
    size_t nh = 1;
    for (int d : hdims) nh *= d;
    real *ps = nullptr, *pt = nullptr;
    if (hipMalloc((void **)&ps, 5 * nh * sizeof(real)) != hipSuccess || hipMalloc((void **)&pt, (size_t)nt * nh * sizeof(real)) != hipSuccess) endrun("hipMalloc");
    real5d state(ps, {5, hdims[0], hdims[1], hdims[2], hdims[3]}), tracers(pt, {nt, hdims[0], hdims[1], hdims[2], hdims[3]});
    dycore.convert_coupler_to_dynamics(coupler, state, tracers);
    for (auto &n : names) (void)hipMemsetAsync(dm.get<real, 4>(n).data(), 0xFF, ncell * sizeof(real), 0);   // NaN bit patterns
    dycore.convert_dynamics_to_coupler(coupler, realConst5d(ps, state.dims()), realConst5d(pt, tracers.dims()));
    if (hipDeviceSynchronize() != hipSuccess) endrun("device error");
    (void)hipFree(ps); (void)hipFree(pt);
  }
#endif
  auto one_step = [&]() {
#ifdef PAMC_DYCORE
    coupler.run_module("dycore", [&](pam::PamCoupler &c) { dycore.timeStep(c); });      // driver.cpp:248
#else
    if (world == 1) {
      coupler.run_module("dycore", [&](pam::PamCoupler &c) { dycore.timeStep(c); });    // driver.cpp:248
    } else {
      // the dynamics step of the WHOLE ensemble (awfl/Dycore.h:141-145 takes the minimum over every member): this device's
      // minimum, then the minimum over the ranks' N host doubles
      coupler.run_module("dycore", [&](pam::PamCoupler &c) {
        const real dt_all = hmin(rank, dycore.compute_time_step(c));
        dycore.timeStep(c, dt_all);
      });
    }
    bench.substeps += dycore.last_ncycles();
#endif
    if (J.with_sponge) coupler.run_module("sponge_layer", modules::sponge_layer);       // driver.cpp:250
    if (J.with_micro) {                                                                  // driver.cpp:253
      if (world == 1) {
        coupler.run_module("micro", [&](pam::PamCoupler &c) { micro.timeStep(c); });
      } else {
        // Kessler's sedimentation sub-cycle count is a reduction over ALL members as well (rainsplit = ceil(dt / minval(dt2d)),
        // physics/micro/kessler/Microphysics.h:385-390): this shard's stable step, the minimum over the ranks, one count for all
        coupler.run_module("micro", [&](pam::PamCoupler &c) {
          const real dt_max_all = hmin(rank, micro.max_stable_dt(c));
          const real crm_dt = c.get_option<real>("crm_dt");
          const int rainsplit = std::max(1, (int)std::ceil(crm_dt / dt_max_all));
          micro.timeStep(c, rainsplit);
        });
      }
    }
  };
  if (J.bench_steps > 0) {
    for (int s = 0; s < J.bench_warmup; s++) one_step();
    if (hipDeviceSynchronize() != hipSuccess) endrun("device error");
    hmin.barrier(rank);
    bench.substeps = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int s = 0; s < J.bench_steps; s++) one_step();
    if (hipDeviceSynchronize() != hipSuccess) endrun("device error");
    hmin.barrier(rank);
    bench.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  } else {
    for (int s = 0; s < J.nsteps; s++) one_step();
  }
  if (hipDeviceSynchronize() != hipSuccess) endrun("device error");
  for (size_t f = 0; f < names.size() && J.want_output; f++) {
    if (hipMemcpy(buf.data(), dm.get<real, 4>(names[f]).data(), ncell * sizeof(real), hipMemcpyDeviceToHost) != hipSuccess) endrun("memcpy");
    real *dst = J.fields[f].data();
    for (size_t c = 0; c < ncol; c++) std::memcpy(&dst[c * J.nens + lo], &buf[c * ne], ne * sizeof(real));
  }
  if (J.with_micro && J.want_output) {
    const size_t n2 = (size_t)ny * nx;
    if (hipMemcpy(buf.data(), dm.get<real, 3>("precl").data(), n2 * ne * sizeof(real), hipMemcpyDeviceToHost) != hipSuccess) endrun("memcpy");
    for (size_t c = 0; c < n2; c++) std::memcpy(&J.precl[c * J.nens + lo], &buf[c * ne], ne * sizeof(real));
  }
  dycore.finalize(coupler);                                                // driver.cpp:285
}

// ------------------------------------------------------------------------------------------------------------------------------
//   driver --yaml <input.yaml> [--nens N] [--steps S] [--check] <output.bin | ->
// The reference driver's OWN flow from its own kind of input file (standalone/mmf_simplified/driver.cpp:79-297; the flat
// `key : value` YAML files under standalone/mmf_simplified/inputs/): sim_time, crm_nx, crm_ny, nens, xlen, ylen, dt_gcm, dt_crm_phys,
// out_freq, vcoords [, crm_nz, zlen, idealized, apply_sponge, initData].  What runs, in the reference's order:
//   allocate_coupler_state -> set_grid -> micro.init -> dycore.init [-> initData on the device when `idealized`] ->
//   initialize_from_supercell_column (driver.cpp:19-77: supercell_init column -> gcm_* columns -> broadcast_initial_gcm_column ->
//   perturb_temperature, all on the device) when not idealized -> per GCM step { declare_current_profile_as_hydrostatic (what E3SM's
//   MMF driver does once per GCM step; the standalone reference never calls it and runs on uninitialised variable_gravity);
//   per CRM step { dycore -> sponge_layer -> micro } }.
// Not run: P3 and SHOC (the CI build's micro / sgs; external SCREAM code, out of scope) -- Kessler stands in as the microphysics -- and
// modules::*_gcm_forcing_tendencies, whose field list is P3's tracer set (pam_core/modules/gcm_forcing.h:33-42).
// vcoords: "uniform" (driver.cpp:135-153, with crm_nz and zlen) or the reference's file name `vcoords_equal_<N>_<H>km.nc` -- netCDF-4,
// unreadable here; its contents are what the name says (read from the raw bytes of the reference's copy: 51 interfaces 0, 400, ...,
// 20000 m for vcoords_equal_50_20km.nc): N equal levels up to H km.
// --nens overrides the file's ensemble size, --steps stops after S CRM steps, --check switches the dycore's conservation check on
// (Dycore.h:224-251) and the last stdout line is a JSON object with the run's statistics.
static std::map<std::string, std::string> read_flat_yaml(const std::string &file) {
  std::ifstream in(file);
  if (!in) die("cannot open the YAML input");
  std::map<std::string, std::string> kv;
  std::string line;
  auto trim = [](std::string v) {
    auto a = v.find_first_not_of(" \t\r\""), b = v.find_last_not_of(" \t\r\"");
    return a == std::string::npos ? std::string() : v.substr(a, b - a + 1);
  };
  while (std::getline(in, line)) {
    auto hash = line.find('#');
    if (hash != std::string::npos) line.erase(hash);
    auto colon = line.find(':');
    if (colon == std::string::npos) continue;
    kv[trim(line.substr(0, colon))] = trim(line.substr(colon + 1));
  }
  return kv;
}

// driver.cpp:19-77
static void initialize_from_supercell_column(std::vector<real> const &zint_in, pam::PamCoupler &coupler) {
  const int nz = coupler.get_nz(), nens = coupler.get_nens();
  auto &dm = coupler.get_data_manager_device_readwrite();
  const real R_d = coupler.get_option<real>("R_d"), R_v = coupler.get_option<real>("R_v"), grav = coupler.get_option<real>("grav");
  double *zdev = nullptr, *cols = nullptr;
  if (hipMalloc((void **)&zdev, (nz + 1) * sizeof(double)) != hipSuccess || hipMalloc((void **)&cols, (size_t)6 * nz * sizeof(double)) != hipSuccess) endrun("hipMalloc");
  if (hipMemcpy(zdev, zint_in.data(), (nz + 1) * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) endrun("memcpy");
  // rho_d, uvel, vvel, wvel, temp, rho_v columns (supercell_init.h:7-135)
  if (pam_amd_supercell_init(nz, zdev, R_d, R_v, grav, cols, cols + nz, cols + 2 * nz, cols + 3 * nz, cols + 4 * nz, cols + 5 * nz, nullptr))
    endrun(pam_amd_awfl_last_error());
  std::vector<double> h((size_t)6 * nz), col((size_t)nz * nens);
  if (hipMemcpy(h.data(), cols, h.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) endrun("memcpy");
  auto put = [&](char const *name, int f) {                                  // driver.cpp:55-69: every member gets the column
    for (int k = 0; k < nz; k++)
      for (int e = 0; e < nens; e++) col[(size_t)k * nens + e] = f < 0 ? 0.0 : h[(size_t)f * nz + k];
    if (hipMemcpy(dm.get<real, 2>(name).data(), col.data(), col.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) endrun("memcpy");
  };
  put("gcm_density_dry", 0); put("gcm_uvel", 1); put("gcm_vvel", 2); put("gcm_wvel", 3); put("gcm_temp", 4); put("gcm_water_vapor", 5);
  put("ref_density_dry", 0); put("ref_density_vapor", 5); put("ref_density_liq", -1); put("ref_density_ice", -1); put("ref_temp", 4);
  (void)hipFree(zdev); (void)hipFree(cols);
  modules::broadcast_initial_gcm_column(coupler);                            // driver.cpp:72
  int *seeds = nullptr;                                                      // driver.cpp:74-76: int1d seeds("seeds", nens); seeds = 0
  if (hipMalloc((void **)&seeds, nens * sizeof(int)) != hipSuccess || hipMemset(seeds, 0, nens * sizeof(int)) != hipSuccess) endrun("hipMalloc");
  modules::perturb_temperature(coupler, intConst1d(seeds, {nens}));
  if (hipDeviceSynchronize() != hipSuccess) endrun("device error");
  (void)hipFree(seeds);
}

#ifndef PAMC_DYCORE
struct YamlDebug { bool no_micro = false, no_sponge = false, sync = false; };      // bisection switches (tools/repro_ci_run.py has the Python twin)
static int run_yaml(const std::string &file, int nens_override, int steps_limit, bool check, const std::string &outfile, YamlDebug dbg = YamlDebug()) {
  auto kv = read_flat_yaml(file);
  auto has = [&](char const *k) { return kv.count(k) > 0; };
  auto num = [&](char const *k, double dflt, bool required = false) {
    if (!has(k)) { if (required) die((std::string("YAML key missing: ") + k).c_str()); return dflt; }
    return std::atof(kv[k].c_str());
  };
  auto flag = [&](char const *k, bool dflt) { return has(k) ? (kv[k] == "true" || kv[k] == "True" || kv[k] == "1") : dflt; };
  const bool idealized = flag("idealized", false);                           // driver.cpp:91-94
  const bool apply_sponge = flag("apply_sponge", !idealized);
  const bool apply_gcm_forcing = flag("apply_gcm_forcing", !idealized);
  const double sim_time = num("sim_time", 0, true);
  const int crm_nx = (int)num("crm_nx", 0, true), crm_ny = (int)num("crm_ny", 0, true);
  const int nens = nens_override > 0 ? nens_override : (int)num("nens", 0, true);
  const double xlen = num("xlen", -1), ylen = num("ylen", -1), zlen = num("zlen", -1);
  double dt_gcm = num("dt_gcm", sim_time), dt_crm_phys = num("dt_crm_phys", 0, true);
  const double out_freq = num("out_freq", -1);
  if (!has("vcoords")) die("YAML key missing: vcoords");
  const std::string vcoords = kv["vcoords"];
  const int nsteps_gcm = (int)std::ceil(sim_time / dt_gcm);                  // driver.cpp:115-118
  dt_gcm = sim_time / nsteps_gcm;
  const int nsteps_crm_phys = (int)std::ceil(dt_gcm / dt_crm_phys);
  dt_crm_phys = dt_gcm / nsteps_crm_phys;
  std::vector<real> zint;
  if (vcoords == "uniform") {                                                // driver.cpp:135-153
    const int crm_nz = (int)num("crm_nz", 0, true);
    if (!(zlen > 0)) die("vcoords: uniform needs zlen");
    const real dz = zlen / (crm_nz - 1);
    zint.resize(crm_nz + 1);
    for (int k = 0; k <= crm_nz; k++) zint[k] = (k == 0) ? 0 : (k == crm_nz ? zlen : k * dz - dz / 2);
  } else {
    int n = 0; double hkm = 0;
    if (std::sscanf(vcoords.c_str(), "vcoords_equal_%d_%lfkm.nc", &n, &hkm) != 2 || n < 3 || !(hkm > 0))
      die("vcoords: only `uniform` and `vcoords_equal_<N>_<H>km.nc` are understood (netCDF is not available here)");
    zint.resize(n + 1);
    for (int k = 0; k <= n; k++) zint[k] = hkm * 1000.0 * k / n;
  }
  const int crm_nz = (int)zint.size() - 1;
  if (!(xlen > 0) || !(ylen > 0)) die("xlen / ylen must be given");
  if (hipSetDevice(0) != hipSuccess) die("no HIP device");
  int rcode = 0;
  try {
    pam::PamCoupler coupler;
    coupler.set_option<real>("gcm_physics_dt", dt_gcm);                      // driver.cpp:122-123
    coupler.set_option<real>("crm_dt", dt_crm_phys);
    if (idealized) coupler.set_option<std::string>("standalone_input_file", file);   // driver.cpp:125-128
    coupler.allocate_coupler_state(crm_nz, crm_ny, crm_nx, nens);            // driver.cpp:177
    coupler.set_grid(xlen, ylen, zint);                                      // driver.cpp:180
    Dycore dycore;
    Microphysics micro;
    micro.init(coupler);                                                     // driver.cpp:189 (sgs: SHOC, out of scope)
    dycore.init(coupler);                                                    // driver.cpp:191
    std::printf("Dycore: %s\nMicro : %s\nSGS   : none (SHOC is out of scope)\n\n", dycore.dycore_name(), micro.micro_name().c_str());   // driver.cpp:203-205
    std::printf("crm_nx:   %d\ncrm_ny:   %d\ncrm_nz:   %d\nxlen (m): %g\nylen (m): %g\n", crm_nx, crm_ny, crm_nz, xlen, ylen);
    if (apply_gcm_forcing) std::printf("apply_gcm_forcing: not run (modules::gcm_forcing works on P3's tracer set; P3 is out of scope)\n");
    if (!idealized) initialize_from_supercell_column(zint, coupler);        // driver.cpp:220-222
    if (check) dycore.set_debug_conservation(true);
    auto &dm = coupler.get_data_manager_device_readwrite();
    const size_t ncell = (size_t)crm_nz * crm_ny * crm_nx * nens;
    std::vector<real> buf(ncell);
    auto fetch = [&](std::string const &name) {
      if (hipMemcpy(buf.data(), dm.get<real const, 4>(name).data(), ncell * sizeof(real), hipMemcpyDeviceToHost) != hipSuccess) endrun("memcpy");
    };
    double etime_gcm = 0, maxw_all = 0, cons_max_rel = 0;
    int num_out = 0, crm_steps = 0;
    long substeps = 0, cons_violations = 0;
    bool stop = false;
    std::string maxw_series;
    for (int step_gcm = 0; step_gcm < nsteps_gcm && !stop; ++step_gcm) {
      dycore.declare_current_profile_as_hydrostatic(coupler);               // (E3SM's MMF driver: once per GCM step)
      for (int step_crm_phys = 0; step_crm_phys < nsteps_crm_phys && !stop; ++step_crm_phys) {
        coupler.run_module("dycore", [&](pam::PamCoupler &c) { dycore.timeStep(c); });       // driver.cpp:248
        substeps += dycore.last_ncycles();
        if (check) {
          cons_violations += dycore.conservation_violations();
          cons_max_rel = std::max(cons_max_rel, (double)dycore.conservation_max_rel_diff());
        }
        if (dbg.sync && hipDeviceSynchronize() != hipSuccess) endrun("device error");
        if (apply_sponge && !dbg.no_sponge) coupler.run_module("sponge_layer", modules::sponge_layer);         // driver.cpp:249-251
        if (dbg.sync && hipDeviceSynchronize() != hipSuccess) endrun("device error");
        if (!dbg.no_micro) coupler.run_module("micro", [&](pam::PamCoupler &c) { micro.timeStep(c); });         // driver.cpp:253
        if (dbg.sync && hipDeviceSynchronize() != hipSuccess) endrun("device error");
        crm_steps++;
        etime_gcm = step_gcm * dt_gcm + (step_crm_phys + 1) * dt_crm_phys;  // driver.cpp:255
        if (out_freq >= 0. && etime_gcm / out_freq >= num_out + 1) {        // driver.cpp:257-271 (the netCDF output itself: out of scope)
          fetch("wvel");
          double maxw = 0;
          for (real v : buf) maxw = std::max(maxw, std::fabs(v));
          if (!(maxw == maxw)) maxw = INFINITY;
          maxw_all = std::max(maxw_all, maxw);
          std::printf("Etime , dtphys, maxw: %g , %g , %10.6g\n", etime_gcm, dt_crm_phys, maxw);
          char t[64];
          std::snprintf(t, sizeof(t), "%s%.6g", maxw_series.empty() ? "" : ",", maxw);
          maxw_series += t;
          num_out++;
        }
        if (steps_limit > 0 && crm_steps >= steps_limit) stop = true;
      }
    }
    if (hipDeviceSynchronize() != hipSuccess) endrun("device error");
    std::printf("Simulation Time: %g\n", etime_gcm);
    // statistics of the final state + the output file (same layout as the binary mode: the fields, then precl)
    std::vector<std::string> names = {"density_dry", "uvel", "vvel", "wvel", "temp"};
    for (auto &n : coupler.get_tracer_names()) names.push_back(n);
    std::ofstream out;
    if (outfile != "-") out.open(outfile, std::ios::binary);
    bool finite = true;
    double rho_min = INFINITY, t_min = INFINITY, t_max = -INFINITY, w_max = 0, qmin = INFINITY;
    long t_min_at = -1;
    for (size_t f = 0; f < names.size(); f++) {
      fetch(names[f]);
      for (size_t c = 0; c < ncell; c++) {
        const real v = buf[c];
        if (!std::isfinite(v)) finite = false;
        if (f == 0) rho_min = std::min(rho_min, (double)v);
        if (f == 3) w_max = std::max(w_max, (double)std::fabs(v));
        if (f == 4) { if (v < t_min) { t_min = v; t_min_at = (long)c; } t_max = std::max(t_max, (double)v); }
        if (f >= 5) qmin = std::min(qmin, (double)v);
      }
      if (out.is_open()) out.write((char *)buf.data(), ncell * sizeof(real));
    }
    if (out.is_open()) {
      const size_t n2 = (size_t)crm_ny * crm_nx * nens;
      if (hipMemcpy(buf.data(), dm.get<real const, 3>("precl").data(), n2 * sizeof(real), hipMemcpyDeviceToHost) != hipSuccess) endrun("memcpy");
      out.write((char *)buf.data(), n2 * sizeof(real));
    }
#ifdef PAM_FUNCTION_TIMERS
    pam::function_timers::print();
#endif
    std::printf("{\"crm_steps\": %d, \"substeps\": %ld, \"etime\": %.9g, \"dt_crm_phys\": %.9g, \"nens\": %d, \"nx\": %d, \"ny\": %d, \"nz\": %d, "
                "\"finite\": %s, \"rho_d_min\": %.9g, \"temp_min\": %.9g, \"temp_max\": %.9g, \"maxw_final\": %.9g, \"maxw_any_output\": %.9g, "
                "\"temp_min_level\": %ld, \"tracer_min\": %.9g, \"maxw_series\": [%s], \"conservation_checked\": %s, \"conservation_violations\": %ld, "
                "\"conservation_max_rel_diff\": %.6e}\n",
                crm_steps, substeps, etime_gcm, dt_crm_phys, nens, crm_nx, crm_ny, crm_nz, finite ? "true" : "false", rho_min, t_min, t_max, w_max,
                maxw_all, t_min_at < 0 ? -1L : t_min_at / ((long)crm_ny * crm_nx * nens), qmin, maxw_series.c_str(), check ? "true" : "false", cons_violations, cons_max_rel);
    dycore.finalize(coupler);                                                // driver.cpp:285
  } catch (std::string &msg) {
    std::fprintf(stderr, "driver: endrun: %s\n", msg.c_str());
    rcode = 1;
  }
  return rcode;
}
#endif

int main(int argc, char **argv) {
  int gpus = 1, tile = 1, a = 1;
#ifndef PAMC_DYCORE
  if (argc >= 4 && std::string(argv[1]) == "--yaml") {
    int nens_override = 0, steps_limit = 0, b = 3;
    bool check = false;
    YamlDebug dbg;
    for (; b < argc - 1; b++) {
      const std::string o(argv[b]);
      if (o == "--nens" && b + 1 < argc - 1) nens_override = std::atoi(argv[++b]);
      else if (o == "--steps" && b + 1 < argc - 1) steps_limit = std::atoi(argv[++b]);
      else if (o == "--check") check = true;
      else if (o == "--no-micro") dbg.no_micro = true;
      else if (o == "--no-sponge") dbg.no_sponge = true;
      else if (o == "--sync") dbg.sync = true;
      else die("usage: driver --yaml <input.yaml> [--nens N] [--steps S] [--check] <output.bin | ->");
    }
    return run_yaml(argv[2], nens_override, steps_limit, check, argv[argc - 1], dbg);
  }
#endif
  Job J;
  for (; a < argc && argv[a][0] == '-' && argv[a][1] == '-'; a++) {
    const std::string o(argv[a]);
    if (o == "--gpus" && a + 1 < argc) gpus = std::atoi(argv[++a]);
    else if (o == "--tile" && a + 1 < argc) tile = std::atoi(argv[++a]);
    else if (o == "--bench" && a + 2 < argc) { J.bench_steps = std::atoi(argv[++a]); J.bench_warmup = std::atoi(argv[++a]); }
    else die("usage: driver [--gpus N] [--tile R] [--bench K W] <input.bin> <output.bin>");
  }
  if (argc - a != 2 || gpus < 1 || tile < 1) die("usage: driver [--gpus N] [--tile R] [--bench K W] <input.bin> <output.bin>");
  std::ifstream in(argv[a], std::ios::binary);
  if (!in) die("cannot open input");
  int64_t hdr[8];
  in.read((char *)hdr, sizeof(hdr));
  const int nens_in = hdr[0];
  J.nens = nens_in * tile; J.nx = hdr[1]; J.ny = hdr[2]; J.nz = hdr[3]; J.nt = hdr[4]; J.nsteps = hdr[5];
  J.mode_a = (hdr[6] & 1) != 0; J.with_sponge = (hdr[6] & 2) != 0; J.with_micro = (hdr[6] & 4) != 0;
  J.halo_roundtrip = (hdr[6] & 8) != 0; J.has_consts = hdr[7] != 0;
  if (J.with_micro && J.nt != 3) die("the Kessler microphysics registers exactly 3 tracers");
  in.read((char *)J.geo, sizeof(J.geo));
  in.read((char *)J.consts, sizeof(J.consts));
  J.zint.resize(J.nz + 1);
  in.read((char *)J.zint.data(), J.zint.size() * sizeof(real));
  J.flags.resize(2 * J.nt);
  in.read((char *)J.flags.data(), J.flags.size());
  in.read((char *)&J.idWV, sizeof(J.idWV));
  const size_t ncol = (size_t)J.nz * J.ny * J.nx;
  J.nens_in = nens_in;
  J.raw.assign(5 + J.nt, std::vector<real>(ncol * nens_in));
  for (int f = 0; f < 5 + J.nt; f++) {
    in.read((char *)J.raw[f].data(), J.raw[f].size() * sizeof(real));
    if (!in) die("short input file");
  }
  J.want_output = std::string(argv[a + 1]) != "-";        // "-": no output file (bench runs of large ensembles)
  if (J.want_output) {
    J.fields.assign(5 + J.nt, std::vector<real>(ncol * J.nens));
    J.precl.assign((size_t)J.ny * J.nx * J.nens, 0.0);
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) die("no HIP device");
  HostMin hmin(gpus);
  std::vector<BenchResult> bench(gpus);
  std::vector<std::string> errors(gpus);
  std::string name;
  std::vector<std::thread> threads;
  for (int r = 0; r < gpus; r++)
    threads.emplace_back([&, r]() {
      try {
        run_rank(J, r, gpus, ndev, hmin, bench[r], name);
      } catch (std::string &msg) {
        errors[r] = msg.empty() ? "endrun" : msg;
        hmin.fail();
      }
    });
  for (auto &t : threads) t.join();
  for (int r = 0; r < gpus; r++)
    if (!errors[r].empty()) { std::fprintf(stderr, "driver: rank %d: endrun: %s\n", r, errors[r].c_str()); return 1; }
  std::printf("Dycore: %s\n", name.c_str());
  if (J.bench_steps > 0) {
    double sec = 0;
    for (auto &b : bench) sec = std::max(sec, b.seconds);
    std::printf("{\"launcher\": \"cpp\", \"ranks\": %d, \"devices\": %d, \"nens_total\": %d, \"nx\": %d, \"ny\": %d, \"nz\": %d, "
                "\"num_tracers\": %d, \"steps\": %d, \"warmup\": %d, \"seconds\": %.9g, \"substeps\": %ld}\n",
                gpus, ndev, J.nens, J.nx, J.ny, J.nz, J.nt, J.bench_steps, J.bench_warmup, sec, bench[0].substeps);
  }
  if (J.want_output) {
    std::ofstream out(argv[a + 1], std::ios::binary);
    for (auto &f : J.fields) out.write((char *)f.data(), f.size() * sizeof(real));
    if (J.with_micro) out.write((char *)J.precl.data(), J.precl.size() * sizeof(real));
  }
  return 0;
}
