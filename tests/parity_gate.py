"""The multi-step parity gate of every HIP-vs-oracle comparison (tests, the C++ driver tests, __graft_entry__.smoke()).

north_star: prognostic fields within rtol 1e-12 (fp64).  rho_d, T (max-norm and element-wise) and water vapour keep 1e-12 however
many sub-steps were run.  The small, noise-dominated fields (u, v, w, tracers other than water vapour) follow the MEASURED error
curve instead of a flat bound: profiles/r03_error_growth_c1.txt (BASELINE config C1, HIP vs oracle beside the oracle's own response
to ONE ulp of T) has u, v, w at 4.5e-13 ... 6.8e-12 of their maxima over 3 ... 30 sub-steps, i.e. within 1e-12 (1 + nsub/3)
throughout -- that is the flow's own sensitivity.  The gate is 1e-11 (1 + nsub/3): the worst of the oracle cases recorded on
MI355X (profiles/r05_parity_worst.json, r06_parity_worst.json: written under PAM_AMD_PARITY_RECORD) sits at 0.30-0.43 of it, so a 4x regression of the worst
case fails, where round 3's flat 1e-9 let 100x through; a relative perturbation of 1e-10 injected into one field turns a case
red (tests/test_gpu_parity.py).  `factor`: the one exception, the degenerate 3 x 3 x 3 grid (the periodic stencil wraps every line
twice; w is 1e-3 m/s of noise there), recorded at 1.43 and gated at 4."""
import json
import os

import numpy as np

TOL_TIGHT = 1e-12
_RECORD = {}


def tol_noise_fields(nsub, factor=1.0):
    return factor * 1.0e-11 * (1.0 + nsub / 3.0)


def worst_errors(got, exp, names):
    worst = {}
    for k in ("density_dry", "temp", "uvel", "vvel", "wvel"):
        scale = max(np.abs(exp[k]).max(), 1e-300)
        worst[k] = float(np.abs(got[k] - exp[k]).max() / scale)
    for t, n in enumerate(names):
        scale = max(np.abs(exp["tracers"][t]).max(), 1e-300)
        worst[n] = float(np.abs(got["tracers"][t] - exp["tracers"][t]).max() / scale)
    for k in ("density_dry", "temp"):   # bounded away from zero: also ELEMENT-WISE
        worst[k + "_elementwise"] = float(np.abs((got[k] - exp[k]) / exp[k]).max())
    return worst


def compare(got, exp, names, nsub, case=None, factor=1.0):
    """rho_d, T (max-norm and element-wise) and water vapour within 1e-12; the noise-dominated fields within tol_noise_fields(nsub).
    `got` / `exp`: dicts with density_dry, temp, uvel, vvel, wvel (arrays) and tracers (sequence in the order of `names`)."""
    worst = worst_errors(got, exp, names)
    if case is not None:
        _RECORD[case] = dict(worst, nsub=nsub, gate_noise_fields=tol_noise_fields(nsub, factor))
        path = os.environ.get("PAM_AMD_PARITY_RECORD")
        if path:
            old = {}
            if os.path.exists(path):
                try:
                    old = json.load(open(path))
                except Exception:
                    old = {}
            old.update(_RECORD)
            json.dump(old, open(path, "w"), indent=1, sort_keys=True)
    loose = tol_noise_fields(nsub, factor)
    for k, e in worst.items():
        tol = TOL_TIGHT if k.split("_elementwise")[0] in ("density_dry", "temp", "water_vapor") else loose
        assert e <= tol, (k, e, tol, worst)
    return worst
