"""pam_amd: MI355X-native AWFL dycore step behind PAM's Dycore / PamCoupler plug-in surface.

Product path: pam_amd.Dycore -> C ABI (include/pam_amd_awfl.h) -> hand-written HIP kernels (pam_amd/csrc).
No CPU fallback exists; `oracle/` (the CPU restatement used as parity checker) is never imported from here.
"""
from .capi import PamAmdError, LIB_PATH  # noqa: F401
from .coupler import PamCoupler, DataManager, Options  # noqa: F401
from .dycore import Dycore  # noqa: F401
from . import parallel  # noqa: F401
from . import modules  # noqa: F401
from .micro import Microphysics  # noqa: F401
