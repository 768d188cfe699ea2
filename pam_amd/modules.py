"""Coupler modules around the dycore in the CRM step loop, mirroring `namespace modules` of pam_core/modules/
(free functions taking the coupler).  Arithmetic is in libpam_amd_awfl.so (pam_amd/csrc/modules_kernels.hip)."""
import ctypes as C

import torch

from . import capi
from .capi import check


def sponge_layer(coupler):
    """modules::sponge_layer(coupler)  (pam_core/modules/sponge_layer.h:8-95).  Options read exactly as the
    reference: "sponge_num_layers" (default 5), "sponge_time_scale" (default 60), "crm_dt"."""
    lib = capi.load()
    nz, ny, nx, nens = coupler.get_nz(), coupler.get_ny(), coupler.get_nx(), coupler.get_nens()
    num_layers = coupler.get_option("sponge_num_layers") if coupler.option_exists("sponge_num_layers") else 5
    time_scale = coupler.get_option("sponge_time_scale") if coupler.option_exists("sponge_time_scale") else 60.0
    dm = coupler.get_data_manager_device_readwrite()
    names = ["density_dry", "uvel", "vvel", "wvel", "temp"] + coupler.get_tracer_names()
    tens = [dm.get(n) for n in names]
    zint = dm.get("vertical_interface_height", readonly=True)
    zmid = dm.get("vertical_midpoint_height", readonly=True)
    ptrs = (C.c_void_p * len(tens))(*[t.data_ptr() for t in tens])
    with torch.cuda.device(coupler.device):
        check(lib.pam_amd_sponge_layer(nens, nx, ny, nz, len(tens), ptrs, zint.data_ptr(), zmid.data_ptr(),
                                       float(coupler.get_option("crm_dt")), int(num_layers), float(time_scale),
                                       None, torch.cuda.current_stream(coupler.device).cuda_stream))


GCM_FORCING_CRM = ("density_dry", "uvel", "vvel", "temp", "water_vapor", "cloud_water", "ice", "cloud_water_num", "ice_num",
                   "rain_num")
GCM_FORCING_GCM = ("gcm_density_dry", "gcm_uvel", "gcm_vvel", "gcm_temp", "gcm_water_vapor", "gcm_cloud_water", "gcm_cloud_ice",
                   "gcm_num_liq", "gcm_num_ice", "gcm_num_rain")
GCM_FORCING_TEND = (("rho_d", "dry density"), ("uvel", "u-velocity"), ("vvel", "v-velocity"), ("temp", "temperature"),
                    ("qtot", "tot water mix ratio"), ("qv", "vap water mix ratio"), ("ql", "liq water mix ratio"),
                    ("qi", "ice water mix ratio"), ("rho_v", "water vapor density"), ("rho_l", "cloud water density"),
                    ("rho_i", "cloud ice density"), ("nc", "liq number"), ("ni", "ice number"), ("nr", "rain number"))


def _ptr_table(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def compute_gcm_forcing_tendencies(coupler):
    """modules::compute_gcm_forcing_tendencies(coupler)  (pam_core/modules/gcm_forcing.h:17-210): once per GCM step.
    Reads option "gcm_physics_dt"; registers the 14 "gcm_forcing_tend_*" (nz,nens) entries on first use (:132-147)."""
    lib = capi.load()
    nz, ny, nx, nens = coupler.get_nz(), coupler.get_ny(), coupler.get_nx(), coupler.get_nens()
    dm = coupler.get_data_manager_device_readwrite()
    crm = [dm.get(n, readonly=True) for n in GCM_FORCING_CRM]
    gcm = [dm.get(n, readonly=True) for n in GCM_FORCING_GCM[:7]] + [dm.get(n) for n in GCM_FORCING_GCM[7:]]   # :51-53 non-const
    if not dm.entry_exists("gcm_forcing_tend_uvel"):
        for n, d in GCM_FORCING_TEND:
            dm.register_and_allocate("gcm_forcing_tend_" + n, "GCM forcing for " + d, (nz, nens), ("z", "nens"))
    tend = [dm.get("gcm_forcing_tend_" + n) for n, _ in GCM_FORCING_TEND]
    with torch.cuda.device(coupler.device):
        check(lib.pam_amd_gcm_forcing_compute(nens, nx, ny, nz, _ptr_table(crm), _ptr_table(gcm), _ptr_table(tend),
                                              float(coupler.get_option("gcm_physics_dt")),
                                              torch.cuda.current_stream(coupler.device).cuda_stream))


def apply_gcm_forcing_tendencies(coupler):
    """modules::apply_gcm_forcing_tendencies(coupler)  (gcm_forcing.h:297-440): every CRM step.  Options "crm_dt",
    "gcm_physics_dt".  Returns the hole-filling mask (bit s: species s filled; bit 4+s: whole-CRM fallback)."""
    lib = capi.load()
    nz, ny, nx, nens = coupler.get_nz(), coupler.get_ny(), coupler.get_nx(), coupler.get_nens()
    dm = coupler.get_data_manager_device_readwrite()
    crm = [dm.get(n) for n in GCM_FORCING_CRM]
    gcm = [dm.get(n, readonly=True) for n in GCM_FORCING_GCM]
    tend = [dm.get("gcm_forcing_tend_" + n, readonly=n not in ("rho_v", "rho_l", "rho_i")) for n, _ in GCM_FORCING_TEND]
    dz = dm.get("vertical_cell_dz")
    work = torch.empty(6 * nz * nens + 2 * nens + 4, dtype=torch.float64, device=coupler.device)
    mask = C.c_int()
    with torch.cuda.device(coupler.device):
        check(lib.pam_amd_gcm_forcing_apply(nens, nx, ny, nz, _ptr_table(crm), _ptr_table(gcm), _ptr_table(tend), dz.data_ptr(),
                                            float(coupler.get_option("crm_dt")), float(coupler.get_option("gcm_physics_dt")),
                                            work.data_ptr(), torch.cuda.current_stream(coupler.device).cuda_stream,
                                            C.byref(mask)))
    return mask.value   # the call synchronised the stream, so `work` may be dropped


BROADCAST_GCM = ("gcm_density_dry", "gcm_uvel", "gcm_vvel", "gcm_wvel", "gcm_temp", "gcm_water_vapor")
BROADCAST_CRM = ("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor")


def _broadcast(coupler, n):
    lib = capi.load()
    dm = coupler.get_data_manager_device_readwrite()
    gcm = [dm.get(name, readonly=True) for name in BROADCAST_GCM[:n]]
    crm = [dm.get(name) for name in BROADCAST_CRM[:n]]
    with torch.cuda.device(coupler.device):
        check(lib.pam_amd_broadcast_initial_gcm_column(coupler.get_nens(), coupler.get_nx(), coupler.get_ny(), coupler.get_nz(), n,
                                                       _ptr_table(gcm), _ptr_table(crm),
                                                       torch.cuda.current_stream(coupler.device).cuda_stream))


def broadcast_initial_gcm_column(coupler):
    """modules::broadcast_initial_gcm_column(coupler)  (pam_core/modules/broadcast_initial_gcm_column.h:8-41)."""
    _broadcast(coupler, 6)


def broadcast_initial_gcm_column_dry_density(coupler):
    """modules::broadcast_initial_gcm_column_dry_density(coupler)  (broadcast_initial_gcm_column.h:44-62)."""
    _broadcast(coupler, 1)


def perturb_temperature(coupler, ids, magnitude=0.1):
    """modules::perturb_temperature(coupler, id, magnitude)  (pam_core/modules/perturb_temperature.h:10-63); `ids` is one
    integer per member.  splitmix64 stands in for the reference's yakl::Random (see include/pam_amd_modules.h)."""
    lib = capi.load()
    nens = coupler.get_nens()
    ids = torch.as_tensor(ids, dtype=torch.int32, device=coupler.device).contiguous()
    if ids.numel() != nens:
        from .coupler import endrun
        endrun("ERROR: size of id array must be the same as nens")        # perturb_temperature.h:20
    temp = coupler.get_data_manager_device_readwrite().get("temp")
    with torch.cuda.device(coupler.device):
        check(lib.pam_amd_perturb_temperature(nens, coupler.get_nx(), coupler.get_ny(), coupler.get_nz(), temp.data_ptr(),
                                              ids.data_ptr(), float(magnitude),
                                              torch.cuda.current_stream(coupler.device).cuda_stream))
    return ids   # keeps the id array alive until the caller drops it (the launch is asynchronous)


def supercell_init(vert_interface, R_d, R_v, grav):
    """supercell_init(...) of the standalone driver (standalone/mmf_simplified/supercell_init.h:7-135) on the device:
    vert_interface (nz+1,) CUDA tensor -> (rho_d_col, uvel_col, vvel_col, wvel_col, temp_col, rho_v_col), each (nz,)."""
    lib = capi.load()
    z = vert_interface.contiguous()
    if z.dim() != 1 or z.dtype != torch.float64 or not z.is_cuda:
        from .coupler import endrun
        endrun("ERROR: supercell_init: vert_interface must be a 1-D fp64 device array")
    nz = z.numel() - 1
    cols = [torch.empty(nz, dtype=torch.float64, device=z.device) for _ in range(6)]
    with torch.cuda.device(z.device):
        check(lib.pam_amd_supercell_init(nz, z.data_ptr(), float(R_d), float(R_v), float(grav), *[c.data_ptr() for c in cols],
                                         torch.cuda.current_stream(z.device).cuda_stream))
    return tuple(cols)
