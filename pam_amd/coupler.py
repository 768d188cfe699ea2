"""Host-side mirror of the slice of `pam::PamCoupler` / `DataManager` / `Options` the AWFL dycore touches.

Same names, argument meaning and error behaviour as the reference so that tests read like the reference's
driver (standalone/mmf_simplified/driver.cpp:120-191):

    coupler = PamCoupler()
    coupler.set_option("crm_dt", 2.0)                    # Options.h:63-95
    coupler.allocate_coupler_state(nz, ny, nx, nens)     # pam_coupler.h:255-356
    coupler.set_grid(xlen, ylen, zint)                   # pam_coupler.h:163-202
    coupler.add_tracer("water_vapor", "", True, True)    # pam_coupler.h:206-213
    dycore.init(coupler)

Arrays live in HBM as torch.float64 CUDA tensors (torch is only the allocator / stream provider here); the
DataManager hands out the tensors themselves (views, not copies) like `dm.get<T,N>` returns non-owning views
(DataManager.h:285-312).  Errors raise PamAmdError where the reference calls endrun().
"""
import torch

from .capi import PamAmdError


def endrun(msg):
    raise PamAmdError(msg)


class Options:
    """pam_core/Options.h: typed key -> value map."""

    def __init__(self):
        self._d = {}

    def set_option(self, key, value):
        self._d[key] = value

    add_option = set_option

    def get_option(self, key, val_if_absent=None):
        if key not in self._d:
            if val_if_absent is not None:
                return val_if_absent
            endrun(f"ERROR: option {key} not found")        # Options.h get_option -> endrun
        return self._d[key]

    def option_exists(self, key):
        return key in self._d

    def delete_option(self, key):
        self._d.pop(key, None)


class DataManager:
    """pam_core/DataManager.h: name -> (device array, dims, dim names, dirty flag)."""

    def __init__(self, device):
        self.device = device
        self._e = {}
        self._dims = {}

    def _check_dims(self, dims, dim_names):
        for n, d in zip(dim_names, dims):
            if n in self._dims and self._dims[n] != d:
                endrun(f"ERROR: dimension {n} already exists with size {self._dims[n]} != {d}")   # DataManager.h:112-127
            self._dims[n] = d

    def register_and_allocate(self, name, desc, dims, dim_names=None, dtype=torch.float64):
        if name in self._e:
            endrun(f"ERROR: Duplicate entry name {name}")                                           # DataManager.h:98-103
        if dim_names:
            self._check_dims(dims, dim_names)
        t = torch.zeros(tuple(dims), dtype=dtype, device=self.device)
        self._e[name] = dict(data=t, desc=desc, dims=tuple(dims), dirty=False, owned=True)
        return t

    def register_existing(self, name, desc, tensor, dim_names=None):
        if name in self._e:
            endrun(f"ERROR: Duplicate entry name {name}")
        if dim_names:
            self._check_dims(tuple(tensor.shape), dim_names)
        self._e[name] = dict(data=tensor, desc=desc, dims=tuple(tensor.shape), dirty=False, owned=False)

    def unregister_and_deallocate(self, name):
        if name not in self._e:
            endrun(f"ERROR: Could not find entry {name}")                                           # DataManager.h:230-234
        del self._e[name]       # a managed entry's storage is released with its last reference; a borrowed one is not ours

    def entry_exists(self, name):
        return name in self._e

    def get(self, name, readonly=False):
        if name not in self._e:
            endrun(f"ERROR: Could not find entry {name}")                                           # DataManager.h:526-533
        if not readonly:
            self._e[name]["dirty"] = True                                                           # DataManager.h:306
        return self._e[name]["data"]

    def get_dimension_size(self, name):
        return self._dims.get(name, -1)

    def clean_all_entries(self):
        for e in self._e.values():
            e["dirty"] = False

    def get_dirty_entries(self):
        return [k for k, e in self._e.items() if e["dirty"]]

    def finalize(self):
        self._e.clear()
        self._dims.clear()


class PamCoupler:
    """pam_core/pam_coupler.h:13-396, restricted to what the dycore and its harness call."""

    def __init__(self, device="cuda:0"):
        self.device = torch.device(device)
        self.options = Options()
        self.dm = DataManager(self.device)
        self.xlen = -1.0
        self.ylen = -1.0
        self.tracers = []   # (name, desc, positive, adds_mass), registration order

    # ---- options façade (pam_coupler.h:99-136)
    def set_option(self, key, value):
        self.options.set_option(key, value)

    add_option = set_option

    def get_option(self, key, val_if_absent=None):
        return self.options.get_option(key, val_if_absent)

    def option_exists(self, key):
        return self.options.option_exists(key)

    # ---- grid getters (pam_coupler.h:59-96)
    def get_nx(self): return self.dm.get_dimension_size("x")
    def get_ny(self): return self.dm.get_dimension_size("y")
    def get_nz(self): return self.dm.get_dimension_size("z")
    def get_nens(self): return self.dm.get_dimension_size("nens")
    def get_xlen(self): return self.xlen
    def get_ylen(self): return self.ylen
    def get_dx(self): return self.xlen / self.get_nx()
    def get_dy(self): return self.ylen / self.get_ny()
    def get_data_manager_device_readonly(self): return self.dm
    def get_data_manager_device_readwrite(self): return self.dm

    def allocate_coupler_state(self, nz, ny, nx, nens):
        d4, n4 = (nz, ny, nx, nens), ("z", "y", "x", "nens")
        for name, desc in (("density_dry", "dry density"), ("uvel", "x-direction velocity"),
                           ("vvel", "y-direction velocity"), ("wvel", "z-direction velocity"), ("temp", "temperature")):
            self.dm.register_and_allocate(name, desc, d4, n4)
        self.dm.register_and_allocate("vertical_interface_height", "vertical interface height", (nz + 1, nens), ("zp1", "nens"))
        self.dm.register_and_allocate("vertical_cell_dz", "vertical grid spacing", (nz, nens), ("z", "nens"))
        self.dm.register_and_allocate("vertical_midpoint_height", "vertical midpoint height", (nz, nens), ("z", "nens"))
        for name in ("gcm_density_dry", "gcm_uvel", "gcm_vvel", "gcm_wvel", "gcm_temp", "gcm_water_vapor",
                     "gcm_cloud_water", "gcm_cloud_ice", "gcm_num_liq", "gcm_num_ice", "gcm_num_rain"):   # pam_coupler.h:271-281
            self.dm.register_and_allocate(name, "GCM column " + name[4:], (nz, nens), ("z", "nens"))

    def set_grid(self, xlen, ylen, zint_in):
        """zint_in: (nz+1,) broadcast to every member, or (nz+1,nens)  (pam_coupler.h:163-202)."""
        nz, nens = self.get_nz(), self.get_nens()
        z = torch.as_tensor(zint_in, dtype=torch.float64, device=self.device)
        if z.dim() == 1:
            z = z[:, None].expand(nz + 1, nens)
        if tuple(z.shape) != (nz + 1, nens):
            endrun("ERROR: set_grid: vertical interfaces must be (nz+1) or (nz+1,nens)")
        self.xlen, self.ylen = float(xlen), float(ylen)
        self.dm.get("vertical_interface_height").copy_(z)
        self.dm.get("vertical_midpoint_height").copy_(0.5 * (z[:-1] + z[1:]))
        self.dm.get("vertical_cell_dz").copy_(z[1:] - z[:-1])

    # ---- tracer registry (pam_coupler.h:206-251)
    def add_tracer(self, name, desc, positive, adds_mass):
        nz, ny, nx, nens = self.get_nz(), self.get_ny(), self.get_nx(), self.get_nens()
        self.dm.register_and_allocate(name, desc, (nz, ny, nx, nens), ("z", "y", "x", "nens"))
        self.tracers.append((name, desc, bool(positive), bool(adds_mass)))

    def get_num_tracers(self): return len(self.tracers)
    def get_tracer_names(self): return [t[0] for t in self.tracers]

    def get_tracer_info(self, name):
        for n, desc, pos, mass in self.tracers:
            if n == name:
                return desc, True, pos, mass
        return "", False, False, False

    def tracer_exists(self, name):
        return any(t[0] == name for t in self.tracers)

    def run_module(self, name, f):
        """pam_coupler.h:139-160 (function-trace flavour: returns the entries the module obtained non-const)."""
        self.dm.clean_all_entries()
        f(self)
        return self.dm.get_dirty_entries()

    # ---- convenience for tests / bench: load numpy coupler fields into HBM and read them back
    def load_fields(self, fields):
        for k in ("density_dry", "uvel", "vvel", "wvel", "temp"):
            self.dm.get(k).copy_(torch.from_numpy(fields[k]))
        for t, name in enumerate(self.get_tracer_names()):
            self.dm.get(name).copy_(torch.from_numpy(fields["tracers"][t]))

    def dump_fields(self):
        import numpy as np
        out = {k: self.dm.get(k, readonly=True).cpu().numpy() for k in ("density_dry", "uvel", "vvel", "wvel", "temp")}
        out["tracers"] = np.stack([self.dm.get(n, readonly=True).cpu().numpy() for n in self.get_tracer_names()])
        return out
