"""ctypes binding of the C ABI in include/pam_amd_awfl.h (libpam_amd_awfl.so, built by __graft_entry__.build()).

There is no fallback: if the shared library is missing this module raises at load(), and if no HIP device
is present `pam_amd_awfl_init` fails with PAM_AMD_ENOGPU -- the product never routes through a CPU path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpam_amd_awfl.so")
_DP = C.POINTER(C.c_double)
_LIB = None


class Config(C.Structure):
    _fields_ = [("nens", C.c_int), ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int), ("num_tracers", C.c_int),
                ("xlen", C.c_double), ("ylen", C.c_double),
                ("R_d", C.c_double), ("cp_d", C.c_double), ("R_v", C.c_double), ("cp_v", C.c_double),
                ("p0", C.c_double), ("grav", C.c_double),
                ("cv_d", C.c_double), ("gamma_d", C.c_double), ("kappa_d", C.c_double), ("cv_v", C.c_double),
                ("C0", C.c_double),
                ("idWV", C.c_int),
                ("tracer_positive", C.c_char_p), ("tracer_adds_mass", C.c_char_p),
                ("vertical_cell_dz", C.c_void_p), ("stream", C.c_void_p)]


class Fields(C.Structure):
    _fields_ = [("density_dry", C.c_void_p), ("uvel", C.c_void_p), ("vvel", C.c_void_p), ("wvel", C.c_void_p),
                ("temp", C.c_void_p), ("tracers", C.POINTER(C.c_void_p))]


class GcmColumns(C.Structure):
    _fields_ = [("gcm_density_dry", C.c_void_p), ("gcm_temp", C.c_void_p), ("gcm_water_vapor", C.c_void_p),
                ("gcm_cloud_water", C.c_void_p), ("gcm_cloud_ice", C.c_void_p)]


# every symbol include/pam_amd_awfl.h declares (tests/test_capi_symbols.py checks the header against this list)
SYMBOLS = {
    "pam_amd_awfl_abi_version": (C.c_int, []),
    "pam_amd_awfl_last_error": (C.c_char_p, []),
    "pam_amd_awfl_init": (C.c_int, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    "pam_amd_awfl_init_idealized": (C.c_int, [C.c_void_p, C.POINTER(Fields), C.c_char_p, C.c_void_p, C.c_void_p]),
    "pam_amd_awfl_finalize": (C.c_int, [C.c_void_p]),
    "pam_amd_awfl_dycore_name": (C.c_char_p, [C.c_void_p]),
    "pam_amd_awfl_get_option": (C.c_int, [C.c_void_p, C.c_char_p, _DP]),
    "pam_amd_awfl_set_balance_hydrostasis_with_gravity": (C.c_int, [C.c_void_p, C.c_int]),
    "pam_amd_awfl_get_array": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                                         C.POINTER(C.c_int)]),
    "pam_amd_awfl_bind_array": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p]),
    "pam_amd_awfl_declare_current_profile_as_hydrostatic": (C.c_int, [C.c_void_p, C.POINTER(Fields),
                                                                     C.POINTER(GcmColumns)]),
    "pam_amd_awfl_compute_time_step": (C.c_int, [C.c_void_p, C.POINTER(Fields), C.c_double, _DP]),
    "pam_amd_awfl_time_step": (C.c_int, [C.c_void_p, C.POINTER(Fields), C.c_double, C.c_double, C.POINTER(C.c_int), _DP]),
    "pam_amd_awfl_convert_coupler_to_dynamics": (C.c_int, [C.c_void_p, C.POINTER(Fields)]),
    "pam_amd_awfl_convert_dynamics_to_coupler": (C.c_int, [C.c_void_p, C.POINTER(Fields)]),
    "pam_amd_awfl_convert_coupler_to_dynamics_arrays": (C.c_int, [C.c_void_p, C.POINTER(Fields), C.c_void_p, C.c_void_p]),
    "pam_amd_awfl_convert_dynamics_to_coupler_arrays": (C.c_int, [C.c_void_p, C.POINTER(Fields), C.c_void_p, C.c_void_p]),
    "pam_amd_awfl_set_kernel_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "pam_amd_awfl_get_kernel_timing": (C.c_int, [C.c_void_p, C.c_char_p, _DP, C.POINTER(C.c_longlong)]),
    "pam_amd_awfl_reset_kernel_timing": (C.c_int, [C.c_void_p]),
    "pam_amd_awfl_set_flux_segment": (C.c_int, [C.c_void_p, C.c_int]),
    "pam_amd_awfl_set_flux_span": (C.c_int, [C.c_void_p, C.c_int]),
    "pam_amd_awfl_set_ensemble_chunks": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "pam_amd_awfl_set_fused_stage": (C.c_int, [C.c_void_p, C.c_int]),
    "pam_amd_awfl_set_range_schedule": (C.c_int, [C.c_void_p, C.c_int]),
    "pam_amd_awfl_set_yz_fold": (C.c_int, [C.c_void_p, C.c_int]),
    "pam_amd_awfl_set_tail_fusion": (C.c_int, [C.c_void_p, C.c_int]),
    "pam_amd_awfl_debug_fail_next_capture": (C.c_int, [C.c_void_p, C.c_int]),
    "pam_amd_awfl_set_debug_conservation": (C.c_int, [C.c_void_p, C.c_int]),
    "pam_amd_awfl_get_conservation": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pam_amd_awfl_conservation_report": (C.c_char_p, [C.c_void_p]),
    "pam_amd_awfl_debug_inject_mass_fault": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]),
    "pam_amd_awfl_set_lane_mapping": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "pam_amd_awfl_set_x_tile": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "pam_amd_awfl_set_x_exchange": (C.c_int, [C.c_void_p, C.c_int]),
    "pam_amd_awfl_set_flux_tile_parts": (C.c_int, [C.c_void_p, C.c_int]),
    "pam_amd_awfl_set_tile_state_parts": (C.c_int, [C.c_void_p, C.c_int]),
    "pam_amd_awfl_set_tracer_grouping": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "pam_amd_awfl_set_flux_tile": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "pam_amd_awfl_set_tile_fusion": (C.c_int, [C.c_void_p, C.c_int]),
    "pam_amd_awfl_set_graph_replay": (C.c_int, [C.c_void_p, C.c_int]),
    "pam_amd_awfl_set_launch_tuning": (C.c_int, [C.c_longlong, C.c_longlong, C.c_longlong]),
    "pam_amd_awfl_set_handle_launch_tuning": (C.c_int, [C.c_void_p, C.c_longlong, C.c_longlong, C.c_longlong]),
    "pam_amd_awfl_get_lane_mapping": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pam_amd_awfl_debug_get_buffer": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "pam_amd_awfl_debug_fct_rows": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_int)]),
    "pam_amd_awfl_debug_flux_stage": (C.c_int, [C.c_void_p, C.c_double]),
    "pam_amd_awfl_debug_weno": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "pam_amd_awfl_debug_pow": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p]),
    "pam_amd_awfl_debug_stage": (C.c_int, [C.c_void_p, C.c_double]),
}

# include/pam_amd_modules.h
MODULE_SYMBOLS = {
    "pam_amd_modules_finalize": (C.c_int, []),
    "pam_amd_sponge_layer": (C.c_int, [C.c_int] * 5 + [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_double, C.c_int,
                                                       C.c_double, C.c_void_p, C.c_void_p]),
    "pam_amd_kessler_time_step": (C.c_int, [C.c_int] * 4 + [C.c_void_p] * 7 + [C.c_double] * 5 + [C.c_void_p, C.c_void_p,
                                                                                                C.c_int, C.POINTER(C.c_int)]),
    "pam_amd_gcm_forcing_compute": (C.c_int, [C.c_int] * 4 + [C.POINTER(C.c_void_p)] * 3 + [C.c_double, C.c_void_p]),
    "pam_amd_gcm_forcing_apply": (C.c_int, [C.c_int] * 4 + [C.POINTER(C.c_void_p)] * 3 + [C.c_void_p, C.c_double, C.c_double,
                                                                                          C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "pam_amd_broadcast_initial_gcm_column": (C.c_int, [C.c_int] * 5 + [C.POINTER(C.c_void_p)] * 2 + [C.c_void_p]),
    "pam_amd_perturb_temperature": (C.c_int, [C.c_int] * 4 + [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]),
    "pam_amd_supercell_init": (C.c_int, [C.c_int, C.c_void_p] + [C.c_double] * 3 + [C.c_void_p] * 6 + [C.c_void_p]),
    "pam_amd_kessler_max_stable_dt": (C.c_int, [C.c_int] * 4 + [C.c_void_p] * 3 + [C.c_double, C.c_void_p, C.c_void_p,
                                                                                   C.POINTER(C.c_double)]),
}


class PamAmdError(RuntimeError):
    """Raised where the reference would call endrun() (pam_core/pam_const.h:249-252)."""


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise PamAmdError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950).  pam_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in list(SYMBOLS.items()) + list(MODULE_SYMBOLS.items()):
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().pam_amd_awfl_last_error()
        raise PamAmdError((msg or b"").decode() + f" [code {rc}]")
