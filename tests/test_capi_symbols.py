"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/pam_amd_awfl.h
declares; argument validation happens before any GPU work; without a GPU the product fails loudly (no fallback)."""
import ctypes as C
import math
import os
import re

import pytest

from pam_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols(header="pam_amd_awfl.h", prefix="pam_amd_awfl_"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(" + prefix + r"\w+)\s*\(", text))


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    declared = _header_symbols()
    assert declared, "no prototypes parsed from the header"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/pam_amd_awfl.h but not exported"
    assert declared == set(capi.SYMBOLS), (declared ^ set(capi.SYMBOLS))
    assert lib.pam_amd_awfl_abi_version() == 5
    mods = _header_symbols("pam_amd_modules.h", "pam_amd_")
    assert mods == set(capi.MODULE_SYMBOLS) and all(hasattr(lib, n) for n in mods)


def test_sponge_layer_rejects_bad_arguments():
    lib = capi.load()
    assert lib.pam_amd_sponge_layer(2, 4, 1, 8, 6, None, None, None, 1.0, 5, 60.0, None, None) == -1
    assert b"sponge_layer" in lib.pam_amd_awfl_last_error()


def test_other_modules_reject_bad_arguments_before_touching_a_device():
    lib = capi.load()
    T10, T14, T6 = (C.c_void_p * 10)(), (C.c_void_p * 14)(), (C.c_void_p * 6)()      # tables of NULL device pointers
    cases = [
        ("kessler", lambda: lib.pam_amd_kessler_time_step(2, 4, 1, 1, None, None, None, None, None, None, None, 1.0, 287., 461.,
                                                          1003., 1e5, None, None, 0, None)),
        ("kessler", lambda: lib.pam_amd_kessler_max_stable_dt(2, 4, 1, 8, None, None, None, 1.0, None, None, None)),
        ("compute_gcm_forcing_tendencies", lambda: lib.pam_amd_gcm_forcing_compute(2, 4, 1, 8, T10, T10, T14, 1200.0, None)),
        ("compute_gcm_forcing_tendencies", lambda: lib.pam_amd_gcm_forcing_compute(2, 4, 1, 8, None, T10, T14, 1200.0, None)),
        ("apply_gcm_forcing_tendencies", lambda: lib.pam_amd_gcm_forcing_apply(2, 4, 1, 0, T10, T10, T14, None, 1.0, 1200.0, None,
                                                                               None, None)),
        ("broadcast_initial_gcm_column", lambda: lib.pam_amd_broadcast_initial_gcm_column(2, 4, 1, 8, 3, T6, T6, None)),
        ("broadcast_initial_gcm_column", lambda: lib.pam_amd_broadcast_initial_gcm_column(2, 4, 1, 8, 6, T6, T6, None)),
        ("perturb_temperature", lambda: lib.pam_amd_perturb_temperature(2, 4, 1, 8, None, None, 0.1, None)),
    ]
    for who, call in cases:
        assert call() == -1, who                                   # PAM_AMD_EINVAL
        assert who.encode() in lib.pam_amd_awfl_last_error(), who


def _cfg(**kw):
    cfg = capi.Config()
    cfg.nens, cfg.nx, cfg.ny, cfg.nz, cfg.num_tracers = 2, 8, 1, 8, 1
    cfg.xlen, cfg.ylen = 8000.0, 8000.0
    for k in ("R_d", "cp_d", "R_v", "cp_v", "p0", "grav"):
        setattr(cfg, k, math.nan)
    cfg.idWV = 0
    cfg.tracer_positive = b"\x01"
    cfg.tracer_adds_mass = b"\x01"
    cfg.vertical_cell_dz = 1   # never dereferenced: validation / device check fail first
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


@pytest.mark.parametrize("bad", [dict(nx=2), dict(ny=2), dict(nz=1), dict(nens=0), dict(num_tracers=0),
                                 dict(num_tracers=51), dict(idWV=3), dict(xlen=-1.0), dict(vertical_cell_dz=None)])
def test_init_rejects_bad_arguments(bad):
    lib = capi.load()
    h = C.c_void_p()
    rc = lib.pam_amd_awfl_init(C.byref(_cfg(**bad)), C.byref(h))
    assert rc == -1 and not h.value            # PAM_AMD_EINVAL, like the reference's endrun()
    assert len(lib.pam_amd_awfl_last_error()) > 0


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = capi.load()
    h = C.c_void_p()
    rc = lib.pam_amd_awfl_init(C.byref(_cfg()), C.byref(h))
    assert rc == -2 and not h.value            # PAM_AMD_ENOGPU
    assert b"no HIP device" in lib.pam_amd_awfl_last_error()


def test_null_handle_calls_fail_cleanly():
    lib = capi.load()
    v = C.c_double()
    assert lib.pam_amd_awfl_get_option(None, b"C0", C.byref(v)) == -1
    assert lib.pam_amd_awfl_time_step(None, None, 1.0, 0.0, None, None) == -1
    assert lib.pam_amd_awfl_finalize(None) == 0


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under pam_amd/ may import, load or link it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pam_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "awfl_oracle" not in text and "libawfl_emu" not in text, os.path.join(dirpath, f)
