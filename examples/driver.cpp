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
#include "physics/micro/kessler_amd/Microphysics.h"

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

int main(int argc, char **argv) {
  int gpus = 1, tile = 1, a = 1;
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
