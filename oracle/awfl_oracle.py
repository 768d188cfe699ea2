"""ctypes binding of the CPU oracle (oracle/awfl_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package `pam_amd` never does.  The class mirrors the reference's `Dycore` call sequence
(dynamics/awfl/Dycore.h: init :835, declare_current_profile_as_hydrostatic :1392,
compute_time_step :65, timeStep :107) over numpy arrays laid out like the coupler's
(`(nz,ny,nx,nens)`, nens fastest, pam_coupler.h:259-263).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_DP = C.POINTER(C.c_double)


def build(out=None, archflags=""):
    """Compile the oracle with gcc (generic arch by default: the .so travels to the GPU box)."""
    args = ["make", "-C", _HERE, "-s"]
    if out:
        args += [f"OUT={out}", f"ARCHFLAGS={archflags}"]
    subprocess.run(args, check=True)
    return out or os.path.join(_HERE, "libawfl_oracle.so")


def _bound_openmp_team():
    """The oracle's loops are `omp parallel for` with the runtime's default team: every CPU the process SEES.  A GPU box shows the whole
    host (hundreds of hardware threads) and lets a job use its share (16 cores for one GPU); with busy neighbours a spinning team of that
    size needs tens of milliseconds per parallel region, and a test of a few thousand regions stops producing output for minutes (round 6:
    one run of tests/test_reference_ci_run.py was killed as hung at the oracle half).  Unless the caller chose otherwise: at most 8 threads,
    sleeping instead of spinning at barriers.  Read by libgomp when it is loaded, so it must happen before the CDLL below."""
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(8, usable))))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    os.environ.setdefault("OMP_DYNAMIC", "false")


def load(path=None):
    global _LIB
    if path is None and _LIB is not None:
        return _LIB
    p = path or os.path.join(_HERE, "libawfl_oracle.so")
    if not os.path.exists(p):
        build()
    _bound_openmp_team()
    lib = C.CDLL(p)
    lib.awfl_oracle_create.restype = C.c_void_p
    lib.awfl_oracle_create.argtypes = [C.c_int] * 5 + [C.c_double] * 2 + [_DP, C.c_char_p, C.c_char_p, C.c_int, _DP]
    lib.awfl_oracle_destroy.argtypes = [C.c_void_p]
    lib.awfl_oracle_set_grav_balance.argtypes = [C.c_void_p, C.c_int]
    lib.awfl_oracle_get_option.restype = C.c_double
    lib.awfl_oracle_get_option.argtypes = [C.c_void_p, C.c_char_p]
    for name in ("variable_gravity", "hy_dens_cells", "hy_pressure_cells", "vert_sten_to_coefs",
                 "vert_weno_recon_lower"):
        f = getattr(lib, "awfl_oracle_" + name)
        f.restype = _DP
        f.argtypes = [C.c_void_p]
    lib.awfl_oracle_halo_elems.restype = C.c_size_t
    lib.awfl_oracle_halo_elems.argtypes = [C.c_void_p]
    lib.awfl_oracle_reconstruct.restype = C.c_double
    lib.awfl_oracle_reconstruct.argtypes = [_DP, C.c_int]
    lib.awfl_oracle_weno_coefs.argtypes = [_DP, _DP]
    lib.awfl_oracle_reconstruct_level.restype = C.c_double
    lib.awfl_oracle_reconstruct_level.argtypes = [C.c_void_p, C.c_int, C.c_int, _DP, C.c_int]
    lib.awfl_oracle_ideal_sigma.argtypes = [_DP, _DP]
    lib.awfl_oracle_variable_matrices.argtypes = [_DP, _DP, _DP]
    lib.awfl_oracle_compute_time_step.restype = C.c_double
    lib.awfl_oracle_compute_time_step.argtypes = [C.c_void_p] + [_DP] * 6 + [C.c_double]
    lib.awfl_oracle_declare_hydrostatic.argtypes = [C.c_void_p] + [_DP] * 6 + [C.POINTER(_DP)]
    lib.awfl_oracle_time_step.restype = C.c_int
    lib.awfl_oracle_time_step.argtypes = [C.c_void_p] + [_DP] * 6 + [C.c_double, C.c_double, _DP]
    lib.awfl_oracle_convert_coupler_to_dynamics.argtypes = [C.c_void_p] + [_DP] * 8
    lib.awfl_oracle_convert_dynamics_to_coupler.argtypes = [C.c_void_p] + [_DP] * 8
    lib.awfl_oracle_compute_tendencies.argtypes = [C.c_void_p] + [_DP] * 4 + [C.c_double]
    lib.awfl_oracle_set_flux_taps.argtypes = [C.c_void_p, _DP, _DP, _DP]
    lib.awfl_oracle_init_thermal.argtypes = [C.c_void_p, _DP] + [_DP] * 6
    lib.awfl_oracle_init_supercell.argtypes = [C.c_void_p, _DP, _DP] + [_DP] * 6
    lib.awfl_oracle_gcm_forcing_compute.argtypes = [C.c_int] * 4 + [C.POINTER(_DP)] * 3 + [C.c_double]
    lib.awfl_oracle_gcm_forcing_apply.restype = C.c_int
    lib.awfl_oracle_gcm_forcing_apply.argtypes = [C.c_int] * 4 + [C.POINTER(_DP)] * 3 + [_DP, C.c_double, C.c_double]
    lib.awfl_oracle_broadcast_gcm_column.argtypes = [C.c_int] * 5 + [C.POINTER(_DP)] * 2
    lib.awfl_oracle_perturb_temperature.argtypes = [C.c_int] * 4 + [_DP, C.POINTER(C.c_int), C.c_double]
    lib.awfl_oracle_supercell_init.argtypes = [C.c_int, _DP] + [C.c_double] * 3 + [_DP] * 6
    lib.awfl_oracle_kessler.restype = C.c_int
    lib.awfl_oracle_kessler.argtypes = [C.c_int] * 4 + [_DP] * 7 + [C.c_double] * 5 + [C.c_int]
    lib.awfl_oracle_sponge_layer.argtypes = [C.c_int] * 5 + [C.POINTER(_DP), _DP, _DP, C.c_double, C.c_int, C.c_double]
    if path is None:
        _LIB = lib
    return lib


def _p(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_DP)


def reconstruct(stencil, ind, lib=None):
    lib = lib or load()
    s = np.ascontiguousarray(stencil, dtype=np.float64)
    return lib.awfl_oracle_reconstruct(_p(s), int(ind))


def weno_coefs(stencil, lib=None):
    lib = lib or load()
    s = np.ascontiguousarray(stencil, dtype=np.float64)
    out = np.zeros(5)
    lib.awfl_oracle_weno_coefs(_p(s), _p(out))
    return out


def ideal_sigma(lib=None):
    lib = lib or load()
    idl = np.zeros(4)
    sig = C.c_double(0)
    lib.awfl_oracle_ideal_sigma(_p(idl), C.byref(sig))
    return idl, sig.value


def variable_matrices(locs, lib=None):
    lib = lib or load()
    l = np.ascontiguousarray(locs, dtype=np.float64)
    s2c = np.zeros((5, 5))
    wrl = np.zeros((3, 3, 3))
    lib.awfl_oracle_variable_matrices(_p(l), _p(s2c), _p(wrl))
    return s2c, wrl


class OracleDycore:
    """Mirror of the reference `Dycore` on numpy coupler fields.

    fields: dict with 'density_dry','uvel','vvel','wvel','temp' (nz,ny,nx,nens) and 'tracers'
    (nt,nz,ny,nx,nens); all float64 C-contiguous; updated in place by time_step().
    """

    def __init__(self, nens, nx, ny, nz, xlen, ylen, dz, tracer_positive, tracer_adds_mass, idWV,
                 consts=None, lib=None):
        self.lib = lib or load()
        self.nens, self.nx, self.ny, self.nz = nens, nx, ny, nz
        self.nt = len(tracer_positive)
        dz = np.ascontiguousarray(np.broadcast_to(np.asarray(dz, dtype=np.float64).reshape(nz, -1), (nz, nens)))
        pos = bytes(bytearray(int(bool(x)) for x in tracer_positive))
        mass = bytes(bytearray(int(bool(x)) for x in tracer_adds_mass))
        cp = None
        if consts is not None:
            self._consts = np.array([consts[k] for k in ("R_d", "cp_d", "R_v", "cp_v", "p0", "grav")], dtype=np.float64)
            cp = _p(self._consts)
        self.h = self.lib.awfl_oracle_create(nens, nx, ny, nz, self.nt, float(xlen), float(ylen), _p(dz), pos, mass,
                                             int(idWV), cp)
        if not self.h:
            raise RuntimeError("awfl_oracle_create failed")

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.awfl_oracle_destroy(self.h)
            self.h = None

    def option(self, key):
        return self.lib.awfl_oracle_get_option(self.h, key.encode())

    def set_grav_balance(self, flag):
        self.lib.awfl_oracle_set_grav_balance(self.h, int(bool(flag)))

    def _view(self, fn, shape):
        ptr = fn(self.h)
        return np.ctypeslib.as_array(ptr, shape=shape)

    @property
    def variable_gravity(self):
        return self._view(self.lib.awfl_oracle_variable_gravity, (self.nz, self.nens))

    @property
    def hy_dens_cells(self):
        return self._view(self.lib.awfl_oracle_hy_dens_cells, (self.nz, self.nens))

    @property
    def hy_pressure_cells(self):
        return self._view(self.lib.awfl_oracle_hy_pressure_cells, (self.nz, self.nens))

    @property
    def vert_sten_to_coefs(self):
        return self._view(self.lib.awfl_oracle_vert_sten_to_coefs, (self.nz + 2, 5, 5, self.nens))

    @property
    def vert_weno_recon_lower(self):
        return self._view(self.lib.awfl_oracle_vert_weno_recon_lower, (self.nz + 2, 3, 3, 3, self.nens))

    def _f(self, fields):
        return [_p(fields[k]) for k in ("density_dry", "uvel", "vvel", "wvel", "temp", "tracers")]

    def declare_current_profile_as_hydrostatic(self, fields, gcm=None):
        if gcm is None:
            self.lib.awfl_oracle_declare_hydrostatic(self.h, *self._f(fields), None)
        else:
            arr = (_DP * 5)(*[_p(gcm[k]) for k in ("gcm_density_dry", "gcm_temp", "gcm_water_vapor",
                                                    "gcm_cloud_water", "gcm_cloud_ice")])
            self.lib.awfl_oracle_declare_hydrostatic(self.h, *self._f(fields), arr)

    def compute_time_step(self, fields, cfl=0.8):
        return self.lib.awfl_oracle_compute_time_step(self.h, *self._f(fields), float(cfl))

    def time_step(self, fields, crm_dt, dt_dyn=0.0):
        out = C.c_double(0)
        n = self.lib.awfl_oracle_time_step(self.h, *self._f(fields), float(crm_dt), float(dt_dyn), C.byref(out))
        return n, out.value

    def init_idealized(self, fields, init_data, zint):
        """Dycore::init's optional idealised data (Dycore.h:986-1090).  zint: (nz+1,) or (nz+1,nens); fills `fields`."""
        zi = np.ascontiguousarray(np.broadcast_to(np.asarray(zint, dtype=np.float64).reshape(self.nz + 1, -1),
                                                  (self.nz + 1, self.nens)))
        zm = np.ascontiguousarray(0.5 * (zi[:-1] + zi[1:]))
        if init_data == "thermal":
            self.lib.awfl_oracle_init_thermal(self.h, _p(zm), *self._f(fields))
        elif init_data == "supercell":
            self.lib.awfl_oracle_init_supercell(self.h, _p(zm), _p(zi), *self._f(fields))
        elif init_data != "external":
            raise ValueError("ERROR: Invalid data_spec")

    def reconstruct_level(self, k, e, stencil, ind):
        """vertical reconstruction with the matrices of index k in [0, nz+1] (Dycore.h:454-469), member e"""
        s = np.ascontiguousarray(stencil, dtype=np.float64)
        return self.lib.awfl_oracle_reconstruct_level(self.h, int(k), int(e), _p(s), int(ind))

    # --- intermediates, for kernel-level parity tests -------------------------------------------
    def halo_shape(self):
        return (self.nz + 6, self.ny + 6, self.nx + 6, self.nens)

    def convert_coupler_to_dynamics(self, fields):
        state = np.full((5,) + self.halo_shape(), np.nan)
        tracers = np.full((self.nt,) + self.halo_shape(), np.nan)
        self.lib.awfl_oracle_convert_coupler_to_dynamics(self.h, *self._f(fields), _p(state), _p(tracers))
        return state, tracers

    def convert_dynamics_to_coupler(self, state, tracers, fields):
        self.lib.awfl_oracle_convert_dynamics_to_coupler(self.h, _p(state), _p(tracers), *self._f(fields))

    def compute_tendencies(self, state, tracers, seed, dt, want_fluxes=False):
        """state/tracers: halo'd, modified in place like the reference.  seed: (nt,nz,ny,nx,nens)."""
        nz, ny, nx, ne, nt = self.nz, self.ny, self.nx, self.nens, self.nt
        st = np.full((5, nz, ny, nx, ne), np.nan)
        tt = np.ascontiguousarray(seed, dtype=np.float64).copy()
        fl = None
        if want_fluxes:
            fl = (np.zeros((5 + nt, nz, ny, nx + 1, ne)), np.zeros((5 + nt, nz, ny + 1, nx, ne)),
                  np.zeros((5 + nt, nz + 1, ny, nx, ne)))
            self.lib.awfl_oracle_set_flux_taps(self.h, _p(fl[0]), _p(fl[1]), _p(fl[2]))
        self.lib.awfl_oracle_compute_tendencies(self.h, _p(state), _p(st), _p(tracers), _p(tt), float(dt))
        if want_fluxes:
            self.lib.awfl_oracle_set_flux_taps(self.h, None, None, None)
            return st, tt, fl
        return st, tt


def supercell_init(zint, consts, lib=None):
    """the standalone driver's supercell column (standalone/mmf_simplified/supercell_init.h:7-135): zint (nz+1,) ->
    (rho_d, uvel, vvel, wvel, temp, rho_v), each (nz,)"""
    lib = lib or load()
    z = np.ascontiguousarray(zint, dtype=np.float64)
    nz = len(z) - 1
    out = [np.zeros(nz) for _ in range(6)]
    lib.awfl_oracle_supercell_init(nz, _p(z), consts["R_d"], consts["R_v"], consts["grav"], *[_p(a) for a in out])
    return out


def sponge_layer(fields, zint, zmid, dt, num_layers=5, time_scale=60.0, lib=None):
    """modules::sponge_layer on numpy coupler fields (dict as for OracleDycore), updated in place.
    zint (nz+1,nens), zmid (nz,nens)."""
    lib = lib or load()
    nz, ny, nx, nens = fields["density_dry"].shape
    arrs = [fields[k] for k in ("density_dry", "uvel", "vvel", "wvel", "temp")] + \
           [fields["tracers"][t] for t in range(fields["tracers"].shape[0])]
    for a in arrs:
        assert a.flags["C_CONTIGUOUS"] and a.dtype == np.float64
    ptrs = (_DP * len(arrs))(*[_p(a) for a in arrs])
    zi = np.ascontiguousarray(zint, dtype=np.float64)
    zm = np.ascontiguousarray(zmid, dtype=np.float64)
    lib.awfl_oracle_sponge_layer(nens, nx, ny, nz, len(arrs), ptrs, _p(zi), _p(zm), float(dt), int(num_layers), float(time_scale))


def kessler(rho_v, rho_c, rho_r, rho_dry, temp, zmid, dt, consts, rainsplit=0, lib=None):
    """Kessler microphysics time step (physics/micro/kessler/Microphysics.h:120-268) on (nz,ny,nx,nens) numpy arrays,
    updated in place.  Returns (precl (ny,nx,nens), rainsplit)."""
    lib = lib or load()
    nz, ny, nx, nens = temp.shape
    precl = np.zeros((ny, nx, nens))
    zm = np.ascontiguousarray(zmid, dtype=np.float64)
    n = lib.awfl_oracle_kessler(nens, nx, ny, nz, _p(rho_v), _p(rho_c), _p(rho_r), _p(rho_dry), _p(temp), _p(precl), _p(zm),
                                float(dt), consts["R_d"], consts["R_v"], consts["cp_d"], consts["p0"], int(rainsplit))
    return precl, n


GCM_FORCING_CRM = ("density_dry", "uvel", "vvel", "temp", "water_vapor", "cloud_water", "ice", "cloud_water_num", "ice_num",
                   "rain_num")
GCM_FORCING_GCM = ("gcm_density_dry", "gcm_uvel", "gcm_vvel", "gcm_temp", "gcm_water_vapor", "gcm_cloud_water", "gcm_cloud_ice",
                   "gcm_num_liq", "gcm_num_ice", "gcm_num_rain")
GCM_FORCING_TEND = tuple("gcm_forcing_tend_" + n for n in ("rho_d", "uvel", "vvel", "temp", "qtot", "qv", "ql", "qi", "rho_v",
                                                           "rho_l", "rho_i", "nc", "ni", "nr"))


def _ptrs(d, names):
    arrs = [d[n] for n in names]
    for a in arrs:
        assert a.flags["C_CONTIGUOUS"] and a.dtype == np.float64
    return (_DP * len(arrs))(*[_p(a) for a in arrs])


def compute_gcm_forcing_tendencies(crm, gcm, dt_gcm, lib=None):
    """modules::compute_gcm_forcing_tendencies (pam_core/modules/gcm_forcing.h:17-210).  crm: dict name -> (nz,ny,nx,nens),
    gcm: dict name -> (nz,nens).  Returns the dict of the 14 gcm_forcing_tend_* arrays (rho_v/l/i still zero)."""
    lib = lib or load()
    nz, ny, nx, nens = crm["density_dry"].shape
    tend = {n: np.zeros((nz, nens)) for n in GCM_FORCING_TEND}
    lib.awfl_oracle_gcm_forcing_compute(nens, nx, ny, nz, _ptrs(crm, GCM_FORCING_CRM), _ptrs(gcm, GCM_FORCING_GCM),
                                        _ptrs(tend, GCM_FORCING_TEND), float(dt_gcm))
    return tend


def apply_gcm_forcing_tendencies(crm, gcm, tend, dz, crm_dt, dt_gcm, lib=None):
    """modules::apply_gcm_forcing_tendencies (gcm_forcing.h:297-440), crm updated in place.  Returns the hole-filling mask."""
    lib = lib or load()
    nz, ny, nx, nens = crm["density_dry"].shape
    dz = np.ascontiguousarray(dz, dtype=np.float64)
    return lib.awfl_oracle_gcm_forcing_apply(nens, nx, ny, nz, _ptrs(crm, GCM_FORCING_CRM), _ptrs(gcm, GCM_FORCING_GCM),
                                             _ptrs(tend, GCM_FORCING_TEND), _p(dz), float(crm_dt), float(dt_gcm))


BROADCAST_GCM = ("gcm_density_dry", "gcm_uvel", "gcm_vvel", "gcm_wvel", "gcm_temp", "gcm_water_vapor")
BROADCAST_CRM = ("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor")


def broadcast_initial_gcm_column(crm, gcm, dry_density_only=False, lib=None):
    """modules::broadcast_initial_gcm_column[_dry_density] (pam_core/modules/broadcast_initial_gcm_column.h)."""
    lib = lib or load()
    nz, ny, nx, nens = crm["density_dry"].shape
    n = 1 if dry_density_only else 6
    lib.awfl_oracle_broadcast_gcm_column(nens, nx, ny, nz, n, _ptrs(gcm, BROADCAST_GCM[:n]), _ptrs(crm, BROADCAST_CRM[:n]))


def perturb_temperature(temp, ids, magnitude=0.1, lib=None):
    """modules::perturb_temperature (pam_core/modules/perturb_temperature.h:10-63) with splitmix64 in place of yakl::Random."""
    lib = lib or load()
    nz, ny, nx, nens = temp.shape
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    assert ids.shape == (nens,) and temp.flags["C_CONTIGUOUS"]
    lib.awfl_oracle_perturb_temperature(nens, nx, ny, nz, _p(temp), ids.ctypes.data_as(C.POINTER(C.c_int)), float(magnitude))
