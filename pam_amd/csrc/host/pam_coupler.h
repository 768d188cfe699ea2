// pam_coupler.h -- minimal, from-scratch work-alike of the slice of PAM's coupler the AWFL dycore uses, backed by
// hipMalloc.  It exists so that the C++ plug-in class (dynamics/awfl_amd/Dycore.h) and a driver can be compiled and run
// in this repository without YAKL; inside PAM the real pam_core/pam_coupler.h is used instead (same member names).
//
// Mirrors, by name and meaning (not by implementation):
//   pam::Options       pam_core/Options.h:63-164      typed key -> value
//   pam::DataManager   pam_core/DataManager.h:90-312   name -> device array + dims, dirty flags, ownership
//   pam::PamCoupler    pam_core/pam_coupler.h:59-293   grid getters, options facade, tracer registry, run_module
//   endrun             pam_core/pam_const.h:249-252    print to stderr and throw
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <iostream>
#include <map>
#include <string>
#include <variant>
#include <vector>

typedef double real;

inline void endrun(std::string msg = "") {
  std::cerr << msg << std::endl;
  throw msg;
}

namespace pam {

// non-owning view of a device array (what dm.get<T,N>() hands out; DataManager.h:285-312)
template <class T>
class DeviceView {
  T *ptr = nullptr;
  std::vector<int> dims_;
 public:
  DeviceView() {}
  DeviceView(T *p, std::vector<int> d) : ptr(p), dims_(std::move(d)) {}
  T *data() const { return ptr; }
  int extent(int i) const { return dims_.at(i); }
  size_t size() const { size_t n = 1; for (int d : dims_) n *= d; return n; }
  std::vector<int> const &dims() const { return dims_; }
};

}  // namespace pam
// pam_const.h:30-55: YAKL device arrays by rank; here every rank is the same non-owning view
typedef pam::DeviceView<real> real5d;
typedef pam::DeviceView<real const> realConst5d;
typedef pam::DeviceView<int> int1d;
typedef pam::DeviceView<int const> intConst1d;
namespace pam {

class Options {
  std::map<std::string, std::variant<int, real, bool, std::string>> opts;
 public:
  template <class T> void set_option(std::string key, T value) { opts[key] = value; }
  template <class T> void add_option(std::string key, T value) { opts[key] = value; }
  bool option_exists(std::string key) const { return opts.count(key) > 0; }
  template <class T> T get_option(std::string key) const {
    auto it = opts.find(key);
    if (it == opts.end()) endrun("ERROR: option " + key + " not found");
    if (!std::holds_alternative<T>(it->second)) endrun("ERROR: option " + key + " requested with the wrong type");
    return std::get<T>(it->second);
  }
  void delete_option(std::string key) { opts.erase(key); }
  void finalize() { opts.clear(); }
};

class DataManager {
  struct Entry {
    void *ptr; std::string desc; std::vector<int> dims; size_t elem; bool owned; bool dirty;
  };
  std::map<std::string, Entry> entries;
  std::map<std::string, int> dimensions;
 public:
  DataManager() {}
  DataManager(DataManager const &) = delete;
  DataManager &operator=(DataManager const &) = delete;
  ~DataManager() { finalize(); }

  template <class T>
  void register_and_allocate(std::string name, std::string desc, std::vector<int> dims,
                             std::vector<std::string> dim_names = std::vector<std::string>(), bool positive = false) {   // DataManager.h:91-95
    if (entries.count(name)) endrun("ERROR: Duplicate entry name " + name);
    for (size_t i = 0; i < dim_names.size(); i++) {
      auto it = dimensions.find(dim_names[i]);
      if (it != dimensions.end() && it->second != dims[i]) endrun("ERROR: dimension " + dim_names[i] + " size mismatch");
      dimensions[dim_names[i]] = dims[i];
    }
    size_t n = 1;
    for (int d : dims) n *= d;
    void *p = nullptr;
    if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) endrun("ERROR: device allocation failed for " + name);
    (void)hipMemset(p, 0, n * sizeof(T));                       // allocate_coupler_state zero-fills (pam_coupler.h:325-355)
    entries[name] = Entry{p, desc, dims, sizeof(T), true, false};
  }

  template <class T>
  void register_existing(std::string name, std::string desc, std::vector<int> dims, T *ptr,
                         std::vector<std::string> dim_names = std::vector<std::string>(), bool positive = false) {       // DataManager.h:158-163
    if (entries.count(name)) endrun("ERROR: Duplicate entry name " + name);
    entries[name] = Entry{(void *)ptr, desc, dims, sizeof(T), false, false};
  }

  // DataManager.h:230-234: frees a managed entry's storage, only forgets a borrowed (register_existing) one
  void unregister_and_deallocate(std::string name) {
    auto it = entries.find(name);
    if (it == entries.end()) endrun("ERROR: Could not find entry " + name);
    if (it->second.owned) (void)hipFree(it->second.ptr);
    entries.erase(it);
  }

  bool entry_exists(std::string name) const { return entries.count(name) > 0; }

  // T const -> read-only access (no dirty flag), T -> read-write (DataManager.h:285-312); N is checked against the rank
  template <class T, int N>
  DeviceView<T> get(std::string name) {
    auto it = entries.find(name);
    if (it == entries.end()) endrun("ERROR: Could not find entry " + name);
    if ((int)it->second.dims.size() != N) endrun("ERROR: rank mismatch for entry " + name);
    if (it->second.elem != sizeof(T)) endrun("ERROR: type mismatch for entry " + name);
    if (!std::is_const<T>::value) it->second.dirty = true;
    return DeviceView<T>((T *)it->second.ptr, it->second.dims);
  }
  template <class T, int N>
  DeviceView<T> get(std::string name) const {
    static_assert(std::is_const<T>::value, "a const DataManager only hands out const data");
    auto it = entries.find(name);
    if (it == entries.end()) endrun("ERROR: Could not find entry " + name);
    if ((int)it->second.dims.size() != N) endrun("ERROR: rank mismatch for entry " + name);
    return DeviceView<T>((T *)it->second.ptr, it->second.dims);
  }

  int get_dimension_size(std::string name) const {
    auto it = dimensions.find(name);
    return it == dimensions.end() ? -1 : it->second;
  }
  void clean_all_entries() { for (auto &e : entries) e.second.dirty = false; }
  std::vector<std::string> get_dirty_entries() const {
    std::vector<std::string> r;
    for (auto &e : entries) if (e.second.dirty) r.push_back(e.first);
    return r;
  }
  void finalize() {
    for (auto &e : entries) if (e.second.owned) (void)hipFree(e.second.ptr);
    entries.clear();
    dimensions.clear();
  }
};

// what yakl::timer_start / yakl::timer_stop are to the reference's run_module under -DPAM_FUNCTION_TIMERS (pam_coupler.h:144-150; YAKL
// prints its timers in yakl::finalize()): wall time per module name, the device drained on both sides
namespace function_timers {
struct Entry { double seconds = 0; long calls = 0; std::chrono::steady_clock::time_point t0; };
inline std::map<std::string, Entry> &table() { static std::map<std::string, Entry> t; return t; }
inline void start(std::string const &name) { (void)hipDeviceSynchronize(); table()[name].t0 = std::chrono::steady_clock::now(); }
inline void stop(std::string const &name) {
  (void)hipDeviceSynchronize();
  auto &e = table()[name];
  e.seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - e.t0).count();
  e.calls++;
}
inline void print(std::ostream &os = std::cout) {
  for (auto &t : table()) os << "timer " << t.first << ": " << t.second.seconds << " s in " << t.second.calls << " calls\n";
}
}  // namespace function_timers

class PamCoupler {
  Options options;
  real xlen = -1, ylen = -1;
  DataManager dm;
  struct Tracer { std::string name, desc; bool positive, adds_mass; };
  std::vector<Tracer> tracers;
 public:
  PamCoupler() {}
  PamCoupler(PamCoupler const &) = delete;
  PamCoupler &operator=(PamCoupler const &) = delete;

  real get_xlen() const { return xlen; }
  real get_ylen() const { return ylen; }
  real get_dx() const { return xlen / get_nx(); }
  real get_dy() const { return ylen / get_ny(); }
  int get_nx() const { return dm.get_dimension_size("x"); }
  int get_ny() const { return dm.get_dimension_size("y"); }
  int get_nz() const { return dm.get_dimension_size("z"); }
  int get_nens() const { return dm.get_dimension_size("nens"); }
  DataManager const &get_data_manager_device_readonly() const { return dm; }
  DataManager &get_data_manager_device_readwrite() { return dm; }

  template <class T> void set_option(std::string key, T value) { options.set_option<T>(key, value); }
  template <class T> void add_option(std::string key, T value) { options.add_option<T>(key, value); }
  template <class T> T get_option(std::string key) const { return options.get_option<T>(key); }
  bool option_exists(std::string key) const { return options.option_exists(key); }

  // pam_coupler.h:139-160: the module runs between the dirty-flag reset / report (-DPAM_FUNCTION_TRACE there; the flags are always
  // kept here) and, with -DPAM_FUNCTION_TIMERS, inside a named timer (yakl::timer_start / timer_stop there: function_timers above)
  template <class F> void run_module(std::string name, F const &f) {
    dm.clean_all_entries();
#ifdef PAM_FUNCTION_TIMERS
    function_timers::start(name);
#endif
    f(*this);
#ifdef PAM_FUNCTION_TIMERS
    function_timers::stop(name);
#endif
#ifdef PAM_FUNCTION_TRACE
    auto dirty = dm.get_dirty_entries();
    std::cout << "MMF Module " << name << " wrote to the following coupler entries: ";
    for (size_t e = 0; e < dirty.size(); e++) std::cout << dirty[e] << (e + 1 < dirty.size() ? ", " : "");
    std::cout << "\n\n";
#endif
  }

  void allocate_coupler_state(int nz, int ny, int nx, int nens) {      // pam_coupler.h:255-293 (the entries the dycore uses)
    for (auto n : {"density_dry", "uvel", "vvel", "wvel", "temp"})
      dm.register_and_allocate<real>(n, "", {nz, ny, nx, nens}, {"z", "y", "x", "nens"});
    dm.register_and_allocate<real>("vertical_interface_height", "", {nz + 1, nens}, {"zp1", "nens"});
    dm.register_and_allocate<real>("vertical_cell_dz", "", {nz, nens}, {"z", "nens"});
    dm.register_and_allocate<real>("vertical_midpoint_height", "", {nz, nens}, {"z", "nens"});
    for (auto n : {"gcm_density_dry", "gcm_uvel", "gcm_vvel", "gcm_wvel", "gcm_temp", "gcm_water_vapor", "gcm_cloud_water",
                   "gcm_cloud_ice", "gcm_num_liq", "gcm_num_ice", "gcm_num_rain",     // pam_coupler.h:270-281
                   "ref_density_dry", "ref_density_vapor", "ref_density_liq", "ref_density_ice", "ref_temp"})
      dm.register_and_allocate<real>(n, "", {nz, nens}, {"z", "nens"});
  }

  // pam_coupler.h:184-202 (one column broadcast to every member); zint_in is a HOST vector of nz+1 interfaces
  void set_grid(real xlen_in, real ylen_in, std::vector<real> const &zint_in) {
    int nz = get_nz(), nens = get_nens();
    if ((int)zint_in.size() != nz + 1) endrun("ERROR: set_grid: need nz+1 interface heights");
    xlen = xlen_in; ylen = ylen_in;
    std::vector<real> zint((size_t)(nz + 1) * nens), zmid((size_t)nz * nens), dz((size_t)nz * nens);
    for (int k = 0; k <= nz; k++)
      for (int e = 0; e < nens; e++) {
        zint[(size_t)k * nens + e] = zint_in[k];
        if (k < nz) {
          zmid[(size_t)k * nens + e] = 0.5 * (zint_in[k] + zint_in[k + 1]);
          dz[(size_t)k * nens + e] = zint_in[k + 1] - zint_in[k];
        }
      }
    (void)hipMemcpy(dm.get<real, 2>("vertical_interface_height").data(), zint.data(), zint.size() * sizeof(real), hipMemcpyHostToDevice);
    (void)hipMemcpy(dm.get<real, 2>("vertical_midpoint_height").data(), zmid.data(), zmid.size() * sizeof(real), hipMemcpyHostToDevice);
    (void)hipMemcpy(dm.get<real, 2>("vertical_cell_dz").data(), dz.data(), dz.size() * sizeof(real), hipMemcpyHostToDevice);
  }

  void add_tracer(std::string name, std::string desc, bool positive, bool adds_mass) {   // pam_coupler.h:206-213
    dm.register_and_allocate<real>(name, desc, {get_nz(), get_ny(), get_nx(), get_nens()}, {"z", "y", "x", "nens"});
    tracers.push_back({name, desc, positive, adds_mass});
  }
  int get_num_tracers() const { return (int)tracers.size(); }
  std::vector<std::string> get_tracer_names() const {
    std::vector<std::string> r;
    for (auto &t : tracers) r.push_back(t.name);
    return r;
  }
  void get_tracer_info(std::string name, std::string &desc, bool &found, bool &positive, bool &adds_mass) const {
    for (auto &t : tracers)
      if (t.name == name) { desc = t.desc; positive = t.positive; adds_mass = t.adds_mass; found = true; return; }
    found = false;
  }
};

}  // namespace pam
