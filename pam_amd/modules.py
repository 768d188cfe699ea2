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
    work = torch.empty(len(tens) * int(num_layers) * nens, dtype=torch.float64, device=coupler.device)
    ptrs = (C.c_void_p * len(tens))(*[t.data_ptr() for t in tens])
    with torch.cuda.device(coupler.device):
        check(lib.pam_amd_sponge_layer(nens, nx, ny, nz, len(tens), ptrs, zint.data_ptr(), zmid.data_ptr(),
                                       float(coupler.get_option("crm_dt")), int(num_layers), float(time_scale),
                                       work.data_ptr(), torch.cuda.current_stream(coupler.device).cuda_stream))
    return work   # keeps the scratch alive until the caller drops it (the launch is asynchronous)
