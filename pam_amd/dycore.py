"""`Dycore`: host-side mirror of the reference plug-in class (dynamics/awfl/Dycore.h) over the C ABI.

Member names, call order and error behaviour follow the reference:

    init(coupler)                                    Dycore.h:835
    declare_current_profile_as_hydrostatic(coupler)  Dycore.h:1392  (the host model calls it once per GCM step)
    compute_time_step(coupler, cfl)                  Dycore.h:65
    timeStep(coupler)                                Dycore.h:107
    convert_coupler_to_dynamics / convert_dynamics_to_coupler   Dycore.h:1336 / :1281
    dycore_name(), finalize(coupler)                 Dycore.h:1544, :1548

All arithmetic happens in libpam_amd_awfl.so (hand-written HIP, gfx950); this file only marshals pointers.
"""
import ctypes as C
import math

import torch

from . import capi
from .capi import PamAmdError, check
from .coupler import endrun


class Dycore:
    def __init__(self):
        self._h = None
        self._lib = None
        self._fields = None
        self._keep = None

    # -------------------------------------------------------------------------------------------- init
    def init(self, coupler, verbose=False):
        lib = capi.load()
        self._lib = lib
        nens, nx, ny, nz = coupler.get_nens(), coupler.get_nx(), coupler.get_ny(), coupler.get_nz()
        if min(nens, nx, ny, nz) < 1:
            endrun("ERROR: coupler state not allocated (allocate_coupler_state) before dycore.init")
        names = coupler.get_tracer_names()
        if "water_vapor" not in names:
            endrun("ERROR: tracer water_vapor must be registered before dycore.init (micro.init runs first)")
        pos = bytes(bytearray(int(coupler.get_tracer_info(n)[2]) for n in names))
        mass = bytes(bytearray(int(coupler.get_tracer_info(n)[3]) for n in names))
        cfg = capi.Config()
        cfg.nens, cfg.nx, cfg.ny, cfg.nz, cfg.num_tracers = nens, nx, ny, nz, len(names)
        cfg.xlen, cfg.ylen = coupler.get_xlen(), coupler.get_ylen()
        for k in ("R_d", "cp_d", "R_v", "cp_v", "p0", "grav",        # Dycore.h:871-876: defaults if absent
                  "cv_d", "gamma_d", "kappa_d", "cv_v", "C0"):       # Dycore.h:883-890: derived if absent
            setattr(cfg, k, float(coupler.get_option(k)) if coupler.option_exists(k) else math.nan)
        cfg.idWV = names.index("water_vapor")
        cfg.tracer_positive, cfg.tracer_adds_mass = pos, mass
        dz = coupler.get_data_manager_device_readonly().get("vertical_cell_dz", readonly=True)
        cfg.vertical_cell_dz = dz.data_ptr()
        cfg.stream = torch.cuda.current_stream(coupler.device).cuda_stream
        h = C.c_void_p()
        with torch.cuda.device(coupler.device):
            check(lib.pam_amd_awfl_init(C.byref(cfg), C.byref(h)))
        self._h = h
        self._device = coupler.device
        # options the reference writes back into the coupler (Dycore.h:866-891,974)
        coupler.set_option("balance_hydrostasis_with_gravity", True)
        for k in ("R_d", "cp_d", "R_v", "cp_v", "p0", "grav", "cv_d", "gamma_d", "kappa_d", "cv_v", "C0"):
            if not coupler.option_exists(k):
                coupler.set_option(k, self.get_option(k))
        coupler.set_option("idWV", cfg.idWV)
        # the dycore's DataManager entries (Dycore.h:868,897-898,975-984): allocated and owned by the DataManager as in the
        # reference (they outlive dycore.finalize); the kernels are bound to that storage
        dm = coupler.get_data_manager_device_readwrite()
        for name in ("variable_gravity", "hy_dens_cells", "hy_pressure_cells", "vert_sten_to_coefs", "vert_weno_recon_lower"):
            t = dm.register_and_allocate(name, "", self._owned_array(name).shape)
            check(lib.pam_amd_awfl_bind_array(self._h, name.encode(), t.data_ptr()))
        dm.register_and_allocate("tracer_adds_mass", "", (len(names),), dtype=torch.bool).copy_(
            torch.tensor(list(mass), dtype=torch.bool))
        dm.register_and_allocate("tracer_positive", "", (len(names),), dtype=torch.bool).copy_(
            torch.tensor(list(pos), dtype=torch.bool))
        # optional idealised initial data (Dycore.h:986-1090; the reference compiles this under PAM_STANDALONE and reads the
        # `initData` key of the YAML file named by option "standalone_input_file")
        if coupler.option_exists("standalone_input_file"):
            import yaml
            with open(coupler.get_option("standalone_input_file")) as fh:
                data_str = str(yaml.safe_load(fh)["initData"])
            self.init_idealized(coupler, data_str)

    def init_idealized(self, coupler, init_data):
        """init_data: "thermal" | "supercell" | "external" (anything else: ERROR: Invalid data_spec, Dycore.h:1002)."""
        self._need()
        f = self._mk_fields(coupler)
        dm = coupler.get_data_manager_device_readonly()
        zmid = dm.get("vertical_midpoint_height", readonly=True)
        zint = dm.get("vertical_interface_height", readonly=True)
        check(self._lib.pam_amd_awfl_init_idealized(self._h, C.byref(f), str(init_data).encode(), zmid.data_ptr(),
                                                    zint.data_ptr()))

    def _owned_array(self, name):
        ptr, dims, nd = C.c_void_p(), (C.c_int * 5)(), C.c_int()
        check(self._lib.pam_amd_awfl_get_array(self._h, name.encode(), C.byref(ptr), dims, C.byref(nd)))
        shape = tuple(dims[i] for i in range(nd.value))
        n = 1
        for d in shape:
            n *= d
        return _device_view(ptr.value, shape, self._device)

    def get_option(self, key):
        v = C.c_double()
        check(self._lib.pam_amd_awfl_get_option(self._h, key.encode(), C.byref(v)))
        return v.value

    # -------------------------------------------------------------------------------------------- helpers
    def _need(self):
        if self._h is None:
            endrun("ERROR: dycore.init(coupler) must be called first")

    def _mk_fields(self, coupler, readonly=False):
        dm = coupler.get_data_manager_device_readonly() if readonly else coupler.get_data_manager_device_readwrite()
        names = coupler.get_tracer_names()
        tens = [dm.get(k, readonly=readonly) for k in ("density_dry", "uvel", "vvel", "wvel", "temp")]
        trc = [dm.get(n, readonly=readonly) for n in names]
        shape = (coupler.get_nz(), coupler.get_ny(), coupler.get_nx(), coupler.get_nens())
        for t in tens + trc:
            if tuple(t.shape) != shape or t.dtype != torch.float64 or not t.is_contiguous() or not t.is_cuda:
                endrun("ERROR: coupler field has wrong shape/type (expected contiguous fp64 (nz,ny,nx,nens) on the GPU)")
        f = capi.Fields()
        f.density_dry, f.uvel, f.vvel, f.wvel, f.temp = [t.data_ptr() for t in tens]
        arr = (C.c_void_p * len(trc))(*[t.data_ptr() for t in trc])
        f.tracers = arr
        self._keep = (tens, trc, arr)
        return f

    def _sync_balance_option(self, coupler):
        # the reference re-reads the option on every call (Dycore.h:284,624,1410)
        want = bool(coupler.get_option("balance_hydrostasis_with_gravity"))
        if want != bool(self.get_option("balance_hydrostasis_with_gravity")):
            check(self._lib.pam_amd_awfl_set_balance_hydrostasis_with_gravity(self._h, int(want)))

    # -------------------------------------------------------------------------------------------- reference surface
    def declare_current_profile_as_hydrostatic(self, coupler, use_gcm_data=False):
        self._need()
        self._sync_balance_option(coupler)
        f = self._mk_fields(coupler, readonly=True)
        g = None
        if use_gcm_data:
            dm = coupler.get_data_manager_device_readonly()
            g = capi.GcmColumns()
            cols = [dm.get(k, readonly=True) for k in ("gcm_density_dry", "gcm_temp", "gcm_water_vapor",
                                                        "gcm_cloud_water", "gcm_cloud_ice")]
            (g.gcm_density_dry, g.gcm_temp, g.gcm_water_vapor, g.gcm_cloud_water, g.gcm_cloud_ice) = \
                [c.data_ptr() for c in cols]
            self._keep_g = cols
        check(self._lib.pam_amd_awfl_declare_current_profile_as_hydrostatic(self._h, C.byref(f),
                                                                            C.byref(g) if g is not None else None))

    def compute_time_step(self, coupler, cfl=0.8):
        self._need()
        f = self._mk_fields(coupler, readonly=True)
        dt = C.c_double()
        check(self._lib.pam_amd_awfl_compute_time_step(self._h, C.byref(f), float(cfl), C.byref(dt)))
        return dt.value

    def timeStep(self, coupler, dt_dyn_hint=0.0):
        """dt_dyn_hint > 0: ensemble-global CFL step agreed between nens shards (see parallel.py); the reference
        has a single process and always derives it locally (Dycore.h:141)."""
        self._need()
        self._sync_balance_option(coupler)
        f = self._mk_fields(coupler)
        n, dt = C.c_int(), C.c_double()
        check(self._lib.pam_amd_awfl_time_step(self._h, C.byref(f), float(coupler.get_option("crm_dt")),
                                               float(dt_dyn_hint), C.byref(n), C.byref(dt)))
        self.last_ncycles, self.last_dt_dyn = n.value, dt.value
        return n.value

    def _halo_arrays(self, coupler, state, tracers):
        hs = 3
        shape = (coupler.get_nz() + 2 * hs, coupler.get_ny() + 2 * hs, coupler.get_nx() + 2 * hs, coupler.get_nens())
        for a, n in ((state, 5), (tracers, coupler.get_num_tracers())):
            if tuple(a.shape) != (n,) + shape or a.dtype != torch.float64 or not a.is_contiguous() or not a.is_cuda:
                endrun("ERROR: state / tracers must be contiguous fp64 (n, nz+6, ny+6, nx+6, nens) arrays on the GPU")

    def convert_coupler_to_dynamics(self, coupler, state=None, tracers=None):
        """Dycore.h:1336.  With the reference's argument list (coupler, state, tracers): fills the interior of the caller's halo'd
        arrays; with the coupler alone: refreshes the dycore's resident state."""
        self._need()
        self._sync_balance_option(coupler)
        f = self._mk_fields(coupler, readonly=True)
        if state is None and tracers is None:
            check(self._lib.pam_amd_awfl_convert_coupler_to_dynamics(self._h, C.byref(f)))
            return
        if state is None or tracers is None:
            endrun("ERROR: convert_coupler_to_dynamics takes (coupler) or (coupler, state, tracers)")
        self._halo_arrays(coupler, state, tracers)
        check(self._lib.pam_amd_awfl_convert_coupler_to_dynamics_arrays(self._h, C.byref(f), state.data_ptr(), tracers.data_ptr()))

    def convert_dynamics_to_coupler(self, coupler, state=None, tracers=None):
        """Dycore.h:1281.  (coupler, state, tracers): coupler fields from the caller's halo'd arrays; coupler alone: from the
        dycore's resident state."""
        self._need()
        f = self._mk_fields(coupler)
        if state is None and tracers is None:
            check(self._lib.pam_amd_awfl_convert_dynamics_to_coupler(self._h, C.byref(f)))
            return
        if state is None or tracers is None:
            endrun("ERROR: convert_dynamics_to_coupler takes (coupler) or (coupler, state, tracers)")
        self._halo_arrays(coupler, state, tracers)
        check(self._lib.pam_amd_awfl_convert_dynamics_to_coupler_arrays(self._h, C.byref(f), state.data_ptr(), tracers.data_ptr()))

    def dycore_name(self):
        lib = self._lib or capi.load()
        return lib.pam_amd_awfl_dycore_name(self._h).decode()

    def finalize(self, coupler=None):
        # Dycore.h:1548: the reference's finalize is empty and takes the coupler const -- the DataManager keeps the dycore's
        # entries (it owns them); only the handle (scratch, streams) is released here
        if self._h is not None:
            check(self._lib.pam_amd_awfl_finalize(self._h))
            self._h = None

    def __del__(self):
        try:
            self.finalize()
        except Exception:
            pass

    # -------------------------------------------------------------------------------------------- measurement / tests
    def set_kernel_timing(self, enable):
        check(self._lib.pam_amd_awfl_set_kernel_timing(self._h, int(bool(enable))))

    def reset_kernel_timing(self):
        check(self._lib.pam_amd_awfl_reset_kernel_timing(self._h))

    def get_kernel_timing(self, name):
        ms, n = C.c_double(), C.c_longlong()
        check(self._lib.pam_amd_awfl_get_kernel_timing(self._h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def set_flux_segment(self, faces):
        check(self._lib.pam_amd_awfl_set_flux_segment(self._h, int(faces)))

    def set_flux_span(self, faces):
        check(self._lib.pam_amd_awfl_set_flux_span(self._h, int(faces)))

    def set_ensemble_chunks(self, chunks, flux_lds_floor_bytes=0):
        check(self._lib.pam_amd_awfl_set_ensemble_chunks(self._h, int(chunks), int(flux_lds_floor_bytes)))

    def set_range_schedule(self, independent):
        check(self._lib.pam_amd_awfl_set_range_schedule(self._h, int(bool(independent))))

    def set_debug_conservation(self, enable=True):
        """the reference's PAM_DEBUG mass check (Dycore.h:36-58, :136-138, :224-251) as an opt-in: masses of every variable and member
        before / after each timeStep by a device reduction"""
        check(self._lib.pam_amd_awfl_set_debug_conservation(self._h, int(bool(enable))))

    def conservation(self):
        """(violations, max_rel_diff, worst_variable, worst_member, report) of the most recent timeStep; variables: tracers in
        registration order, then rho, then rho*theta"""
        n, wv, wm, mr = C.c_int(), C.c_int(), C.c_int(), C.c_double()
        check(self._lib.pam_amd_awfl_get_conservation(self._h, C.byref(n), C.byref(mr), C.byref(wv), C.byref(wm)))
        return n.value, mr.value, wv.value, wm.value, (self._lib.pam_amd_awfl_conservation_report(self._h) or b"").decode()

    def debug_inject_mass_fault(self, variable, k, j, i, member, factor):
        check(self._lib.pam_amd_awfl_debug_inject_mass_fault(self._h, int(variable), int(k), int(j), int(i), int(member), float(factor)))

    def debug_fail_next_capture(self, which):
        check(self._lib.pam_amd_awfl_debug_fail_next_capture(self._h, int(which)))

    def set_tail_fusion(self, mode="auto"):
        """NT > 1, member-lane sweeps: tracer phase 2 + pressure pass + vapour fix-up as one launch ("on" / "auto") or three ("off"); same bits"""
        check(self._lib.pam_amd_awfl_set_tail_fusion(self._h, {"auto": 0, "off": 1, "on": 2}[mode]))

    FOLD = {"auto": 0, "off": 1, "on": 2}

    def set_yz_fold(self, mode="auto"):
        """3-D member-lane stage: the z sweep stores the y+z part of the state's divergence ("on" / "auto") or its own differences
        ("off": the x-sweep loads both); same bits (include/pam_amd_awfl.h)"""
        check(self._lib.pam_amd_awfl_set_yz_fold(self._h, self.FOLD[mode]))

    def set_launch_tuning(self, want_units=0, two_phase_below=-1, split_below=-1):
        """launch-shape thresholds of THIS handle's sweep kernels, in wavefronts (0 / -1 = leave as it is); same results"""
        check(self._lib.pam_amd_awfl_set_handle_launch_tuning(self._h, int(want_units), int(two_phase_below), int(split_below)))

    def set_fused_stage(self, enable):
        check(self._lib.pam_amd_awfl_set_fused_stage(self._h, int(bool(enable))))

    LANES = {"auto": 0, "member": 1, "flat": 2}
    XKERNELS = {"auto": 0, "sweep": 1, "tile": 2}

    def set_lane_mapping(self, yz_lanes="auto", x_kernels="auto"):
        """lane mapping of the fused stage (include/pam_amd_awfl.h): yz_lanes "member" | "flat", x_kernels "sweep" | "tile";
        "auto" = flat / tile for ensembles of fewer than 64 members.  Results do not depend on it."""
        check(self._lib.pam_amd_awfl_set_lane_mapping(self._h, self.LANES[yz_lanes], self.XKERNELS[x_kernels]))

    def set_x_tile(self, row_lanes=0, cells_per_tile=0, lines_per_group=0):
        check(self._lib.pam_amd_awfl_set_x_tile(self._h, int(row_lanes), int(cells_per_tile), int(lines_per_group)))

    def set_tracer_grouping(self, tracers_per_wavefront=0, prefetch=False):
        """separately launched x sweeps of the further tracers: 0 (automatic), 1, 2 or 4 tracers per wavefront; phase 2 with the next
        trip's loads requested one trip ahead (pairs only; experiment); same bits"""
        check(self._lib.pam_amd_awfl_set_tracer_grouping(self._h, int(tracers_per_wavefront), int(bool(prefetch))))

    def set_x_exchange(self, mode="auto"):
        """x tile kernels: neighbouring cells exchange values through "lds" (+ workgroup barriers) or by wavefront "shuffle"s (a whole
        line inside one wavefront) | "auto" (shuffles wherever possible); same bits"""
        check(self._lib.pam_amd_awfl_set_x_exchange(self._h, {"auto": 0, "lds": 1, "shuffle": 2}[mode]))

    def set_flux_tile(self, mode="auto", cells_per_y_tile=0, levels_per_z_tile=0):
        """y/z fluxes of a flat-lane stage: "sweep" (flat-lane sweeps) | "tile" (one tile kernel) | "auto" """
        check(self._lib.pam_amd_awfl_set_flux_tile(self._h, {"auto": 0, "sweep": 1, "tile": 2}[mode], int(cells_per_y_tile),
                                                   int(levels_per_z_tile)))

    def set_tile_state_parts(self, mode="auto"):
        """fused x tile kernel: the state pass of a cell by "one" lane | in three "parts" beside each other | "auto" """
        check(self._lib.pam_amd_awfl_set_tile_state_parts(self._h, {"auto": 0, "one": 1, "parts": 2}[mode]))

    def set_flux_tile_parts(self, mode="auto"):
        """y/z flux tile kernel: the parts of a tile "behind" each other in one workgroup | "beside" each other in workgroups of their own | "auto" """
        check(self._lib.pam_amd_awfl_set_flux_tile_parts(self._h, {"auto": 0, "behind": 1, "beside": 2}[mode]))

    def set_tile_fusion(self, mode="auto"):
        """x tile kernels: the pressure pass and tracer phase 1 "inside" the tile kernel (phase 1 behind the state pass) | "beside" (inside the
        launch, phase 1 in workgroups of its own) | "separate" launches | "auto" """
        check(self._lib.pam_amd_awfl_set_tile_fusion(self._h, {"auto": 0, "separate": 1, "inside": 2, "beside": 3}[mode]))

    def set_graph_replay(self, mode="auto"):
        """a whole timeStep replayed from a captured HIP graph: "on" | "off" | "auto" (launch-bound ensembles)"""
        check(self._lib.pam_amd_awfl_set_graph_replay(self._h, {"auto": 0, "off": 1, "on": 2}[mode]))

    def get_lane_mapping(self):
        flat, tile, cells = C.c_int(), C.c_int(), C.c_int()
        g = (C.c_int * 6)()
        check(self._lib.pam_amd_awfl_get_lane_mapping(self._h, C.byref(flat), C.byref(tile), C.byref(cells), g))
        return {"yz_flat": bool(flat.value), "yz_tile_kernel": flat.value == 2, "x_tiles": bool(tile.value), "x_shuffles": tile.value == 2,
                "flat_cells": bool(cells.value),
                "tile": dict(zip(("W", "nmb", "tc", "halo", "ntl", "lpb"), list(g)))}

    def debug_buffer(self, name):
        ptr, n = C.c_void_p(), C.c_size_t()
        check(self._lib.pam_amd_awfl_debug_get_buffer(self._h, name.encode(), C.byref(ptr), C.byref(n)))
        return _device_view(ptr.value, (n.value,), self._device)

    def debug_fct_rows(self):
        """(rows flagged by the most recent stage's limiter, rows in all, the "some row was flagged" word) -- test hook"""
        n, tot, anyw = C.c_longlong(), C.c_longlong(), C.c_int()
        check(self._lib.pam_amd_awfl_debug_fct_rows(self._h, C.byref(n), C.byref(tot), C.byref(anyw)))
        return n.value, tot.value, bool(anyw.value)

    def debug_weno(self, stencils, level=-1):
        """stencils: (n,5) float64 CUDA tensor -> (left, right) edge values (test hook)"""
        st = stencils.contiguous()
        left, right = torch.empty(st.shape[0], dtype=torch.float64, device=st.device), torch.empty(
            st.shape[0], dtype=torch.float64, device=st.device)
        check(self._lib.pam_amd_awfl_debug_weno(self._h, int(level), st.data_ptr(), st.shape[0], left.data_ptr(), right.data_ptr()))
        return left, right

    def debug_pow(self, x, y):
        """x: float64 CUDA tensor (> 0) -> x ** y as the kernels compute it (test hook)"""
        xx = x.contiguous()
        out = torch.empty_like(xx)
        check(self._lib.pam_amd_awfl_debug_pow(self._h, xx.data_ptr(), xx.numel(), float(y), out.data_ptr()))
        return out

    def debug_stage(self, dt_dyn):
        check(self._lib.pam_amd_awfl_debug_stage(self._h, float(dt_dyn)))

    def debug_flux_stage(self, dt):
        check(self._lib.pam_amd_awfl_debug_flux_stage(self._h, float(dt)))


class _CudaArrayView:
    """__cuda_array_interface__ wrapper so torch can view a raw device pointer owned by the dycore (no copy)."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 3, "strides": None}


def _device_view(ptr, shape, device):
    n = 1
    for d in shape:
        n *= d
    if n == 0 or not ptr:
        return torch.zeros(shape, dtype=torch.float64, device=device)
    with torch.cuda.device(device):
        return torch.as_tensor(_CudaArrayView(ptr, shape), device=device)
