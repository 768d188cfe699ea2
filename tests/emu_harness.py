"""ctypes binding of tests/emu/awfl_emu.cpp (host emulation of the HIP kernel bodies).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "emu", "awfl_emu.cpp")
_SO = os.path.join(_HERE, "emu", "libawfl_emu.so")
_DP = C.POINTER(C.c_double)
_LIB = None


def build():
    deps = [_SRC] + [os.path.join(_HERE, "..", "pam_amd", "csrc", f) for f in
                     ("awfl_device.h", "awfl_vertical.h", "awfl_constants.h")]
    if os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(d) for d in deps):
        return _SO
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", _SO, _SRC], check=True)
    return _SO


def load():
    global _LIB
    if _LIB is None:
        lib = C.CDLL(build())
        lib.emu_init.restype = C.c_void_p
        lib.emu_init.argtypes = [C.c_int] * 5 + [C.c_double] * 2 + [_DP, C.c_int, C.c_char_p, C.c_char_p, _DP, C.c_int]
        lib.emu_destroy.argtypes = [C.c_void_p]
        lib.emu_set_grav_balance.argtypes = [C.c_void_p, C.c_int]
        lib.emu_set_seg.argtypes = [C.c_void_p, C.c_int]
        lib.emu_set_span.argtypes = [C.c_void_p, C.c_int]
        lib.emu_pow.argtypes = [_DP, C.c_int, C.c_double, _DP]
        lib.emu_set_fused.argtypes = [C.c_void_p, C.c_int]
        lib.emu_set_yz_fold.argtypes = [C.c_void_p, C.c_int]
        lib.emu_set_xtr_split.argtypes = [C.c_void_p, C.c_int]
        lib.emu_set_lane_mapping.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.emu_set_x_tile.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        lib.emu_x_tile_geometry.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        lib.emu_set_flux_tile.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        lib.emu_set_tile_pressure.argtypes = [C.c_void_p, C.c_int]
        lib.emu_vz_per_ens.argtypes = [C.c_void_p]
        lib.emu_buffer.restype = _DP
        lib.emu_buffer.argtypes = [C.c_void_p, C.c_char_p]
        lib.emu_declare_hydrostatic.argtypes = [C.c_void_p] + [_DP] * 6 + [C.POINTER(_DP)]
        lib.emu_compute_time_step.restype = C.c_double
        lib.emu_compute_time_step.argtypes = [C.c_void_p] + [_DP] * 6 + [C.c_double]
        lib.emu_convert_coupler_to_dynamics.argtypes = [C.c_void_p] + [_DP] * 6
        lib.emu_flux_stage.argtypes = [C.c_void_p, C.c_double]
        lib.emu_time_step.restype = C.c_int
        lib.emu_time_step.argtypes = [C.c_void_p] + [_DP] * 6 + [C.c_double, C.c_double, _DP]
        _LIB = lib
    return _LIB


def _p(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_DP)


class EmuDycore:
    def __init__(self, nens, nx, ny, nz, xlen, ylen, dz, pos, mass, idWV, consts=None, seg=8):
        self.lib = load()
        self.nens, self.nx, self.ny, self.nz, self.nt = nens, nx, ny, nz, len(pos)
        dz = np.ascontiguousarray(np.broadcast_to(np.asarray(dz, dtype=np.float64).reshape(nz, -1), (nz, nens)))
        cp = None
        if consts is not None:
            self._c = np.array([consts[k] for k in ("R_d", "cp_d", "R_v", "cp_v", "p0", "grav")])
            cp = _p(self._c)
        self.h = self.lib.emu_init(nens, nx, ny, nz, self.nt, float(xlen), float(ylen), cp, int(idWV),
                                   bytes(bytearray(int(bool(x)) for x in pos)),
                                   bytes(bytearray(int(bool(x)) for x in mass)), _p(dz), seg)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.emu_destroy(self.h)
            self.h = None

    def _f(self, f):
        return [_p(f[k]) for k in ("density_dry", "uvel", "vvel", "wvel", "temp", "tracers")]

    def set_grav_balance(self, v):
        self.lib.emu_set_grav_balance(self.h, int(bool(v)))

    def set_seg(self, s):
        self.lib.emu_set_seg(self.h, int(s))

    def set_span(self, s):
        self.lib.emu_set_span(self.h, int(s))

    def set_fused(self, on):
        self.lib.emu_set_fused(self.h, int(bool(on)))

    def set_yz_fold(self, on):
        self.lib.emu_set_yz_fold(self.h, int(bool(on)))

    def set_xtr_split(self, on):
        self.lib.emu_set_xtr_split(self.h, int(bool(on)))

    def set_lane_mapping(self, flat, xtile):
        """fused stage: flat (x, member) lanes in the y/z sweeps and the fix-up; tile kernels in x"""
        self.lib.emu_set_lane_mapping(self.h, int(bool(flat)), int(bool(xtile)))

    def set_x_tile(self, w=0, tc=0, lpb=0):
        self.lib.emu_set_x_tile(self.h, int(w), int(tc), int(lpb))

    def set_flux_tile(self, on, tc_y=0, tc_z=0):
        """with flat lanes: the y/z fluxes as tile kernels (a lane per cell) instead of flat-lane sweeps"""
        self.lib.emu_set_flux_tile(self.h, int(bool(on)), int(tc_y), int(tc_z))

    def set_tile_pressure(self, on):
        self.lib.emu_set_tile_pressure(self.h, int(bool(on)))

    def x_tile_geometry(self):
        g = (C.c_int * 6)()
        self.lib.emu_x_tile_geometry(self.h, g)
        return dict(zip(("W", "nmb", "tc", "halo", "ntl", "lpb"), list(g)))

    @property
    def vz_per_ens(self):
        return bool(self.lib.emu_vz_per_ens(self.h))

    def buffer(self, name, shape):
        return np.ctypeslib.as_array(self.lib.emu_buffer(self.h, name.encode()), shape=shape)

    def declare_current_profile_as_hydrostatic(self, f, gcm=None):
        arr = None
        if gcm is not None:
            arr = (_DP * 5)(*[_p(gcm[k]) for k in ("gcm_density_dry", "gcm_temp", "gcm_water_vapor",
                                                    "gcm_cloud_water", "gcm_cloud_ice")])
        self.lib.emu_declare_hydrostatic(self.h, *self._f(f), arr)

    def compute_time_step(self, f, cfl=0.8):
        return self.lib.emu_compute_time_step(self.h, *self._f(f), float(cfl))

    def convert_coupler_to_dynamics(self, f):
        self.lib.emu_convert_coupler_to_dynamics(self.h, *self._f(f))

    def flux_stage(self, dt):
        self.lib.emu_flux_stage(self.h, float(dt))

    def time_step(self, f, crm_dt, dt_dyn=0.0):
        out = C.c_double(0)
        n = self.lib.emu_time_step(self.h, *self._f(f), float(crm_dt), float(dt_dyn), C.byref(out))
        return n, out.value


def emu_pow(x, y):
    """pow_pos_fast (pam_amd/csrc/awfl_device.h) evaluated on the host"""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    load().emu_pow(_p(x), x.size, float(y), _p(out))
    return out
