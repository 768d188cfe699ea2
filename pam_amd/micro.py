"""Microphysics plug-in mirroring `class Microphysics` of physics/micro/kessler/Microphysics.h (same method names and
coupler side effects).  Arithmetic is in libpam_amd_awfl.so (pam_amd/csrc/modules_kernels.hip); there is no CPU path."""
import ctypes as C
import math

import torch

from . import capi
from .capi import check


class Microphysics:
    """Kessler (1969) warm-rain scheme, three tracers (Microphysics.h:13-38)."""
    ID_V, ID_C, ID_R = 0, 1, 2
    # Microphysics.h:66-71 hard-codes these in the constructor
    R_d, cp_d, cp_v, R_v, p0, grav = 287.0, 1003.0, 1859.0, 461.0, 1.0e5, 9.81

    @staticmethod
    def get_num_tracers():
        return 3

    @staticmethod
    def micro_name():
        return "kessler"

    def init(self, coupler):
        """Microphysics::init (Microphysics.h:55-104): registers the three water tracers (positive, add mass), the
        "precl" diagnostic, zeroes them, and writes the scheme's constants into the coupler options."""
        coupler.add_tracer("water_vapor", "Water Vapor", True, True)
        coupler.add_tracer("cloud_liquid", "Cloud liquid", True, True)
        coupler.add_tracer("precip_liquid", "precip_liquid", True, True)
        dm = coupler.get_data_manager_device_readwrite()
        ny, nx, nens = coupler.get_ny(), coupler.get_nx(), coupler.get_nens()
        dm.register_and_allocate("precl", "precipitation rate", (ny, nx, nens), ("y", "x", "nens"))
        for name in ("water_vapor", "cloud_liquid", "precip_liquid", "precl"):
            dm.get(name).zero_()
        coupler.set_option("micro", "kessler")
        for k in ("R_d", "R_v", "cp_d", "cp_v", "grav", "p0"):
            coupler.set_option(k, getattr(self, k))
        self._work = None

    def _workspace(self, coupler):
        n = coupler.get_nz() * coupler.get_ny() * coupler.get_nx() * coupler.get_nens() + 1
        if getattr(self, "_work", None) is None or self._work.numel() != n or self._work.device != torch.device(coupler.device):
            self._work = torch.empty(n, dtype=torch.float64, device=coupler.device)
        return self._work

    def _arrays(self, coupler):
        dm = coupler.get_data_manager_device_readwrite()
        return (dm.get("water_vapor"), dm.get("cloud_liquid"), dm.get("precip_liquid"), dm.get("density_dry", readonly=True),
                dm.get("temp"), dm.get("precl"), dm.get("vertical_midpoint_height", readonly=True))

    def max_stable_dt(self, coupler):
        """min(0.8 dz / velqr) of kessler() (:377-390) for the current state; ensemble shards all-reduce(MIN) this and pass
        rainsplit = ceil(crm_dt / dt_max) to timeStep."""
        lib = capi.load()
        rv, rc, rr, rd, T, precl, zmid = self._arrays(coupler)
        out = C.c_double()
        with torch.cuda.device(coupler.device):
            check(lib.pam_amd_kessler_max_stable_dt(coupler.get_nens(), coupler.get_nx(), coupler.get_ny(), coupler.get_nz(),
                                                    rr.data_ptr(), rd.data_ptr(), zmid.data_ptr(), float(coupler.get_option("crm_dt")),
                                                    self._workspace(coupler).data_ptr(),
                                                    torch.cuda.current_stream(coupler.device).cuda_stream, C.byref(out)))
        return out.value

    def timeStep(self, coupler, rainsplit=0):
        """Microphysics::timeStep (Microphysics.h:120-268).  Returns the number of sedimentation sub-cycles used."""
        lib = capi.load()
        rv, rc, rr, rd, T, precl, zmid = self._arrays(coupler)
        n = C.c_int()
        with torch.cuda.device(coupler.device):
            check(lib.pam_amd_kessler_time_step(coupler.get_nens(), coupler.get_nx(), coupler.get_ny(), coupler.get_nz(),
                                                rv.data_ptr(), rc.data_ptr(), rr.data_ptr(), rd.data_ptr(), T.data_ptr(),
                                                precl.data_ptr(), zmid.data_ptr(), float(coupler.get_option("crm_dt")),
                                                self.R_d, self.R_v, self.cp_d, self.p0, self._workspace(coupler).data_ptr(),
                                                torch.cuda.current_stream(coupler.device).cuda_stream, int(rainsplit), C.byref(n)))
        return n.value

    def compute_total_mass(self, coupler):
        """Microphysics::compute_total_mass (:108-116): sum of (rho_v+rho_c+rho_r) dz."""
        dm = coupler.get_data_manager_device_readonly()
        zint = dm.get("vertical_interface_height", readonly=True)
        dz = (zint[1:] - zint[:-1])[:, None, None, :]
        return float(((dm.get("water_vapor", readonly=True) + dm.get("cloud_liquid", readonly=True) +
                       dm.get("precip_liquid", readonly=True)) * dz).sum())

    def finalize(self, coupler):
        self._work = None

    @staticmethod
    def rainsplit_for(crm_dt, dt_max):
        return max(1, int(math.ceil(crm_dt / dt_max)))
