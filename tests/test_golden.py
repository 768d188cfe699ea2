"""Committed fixtures under tests/golden/ (see tests/golden/make_golden.py for provenance).

 * survey_kat.json: reference-arithmetic values from SURVEY.md Appendix B -> the oracle must reproduce them
   (bit-exact for the WENO known answers, 1e-11 for the 15-sub-step bubble run).
 * case_*.npz: oracle-generated end-to-end vectors -> (CPU) today's oracle build must reproduce them bit for bit from
   the stored inputs, so the oracle cannot drift silently between rounds; (GPU, -m gpu) the HIP path must match them
   within the parity budget of tests/test_gpu_parity.py.
"""
import importlib.util
import json
import os

import numpy as np
import pytest

from oracle import awfl_oracle as ao
from pam_amd import idealized as idz
from parity_gate import compare

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
mg = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(mg)


def test_survey_known_answers_from_fixture():
    kat = json.load(open(os.path.join(GOLD, "survey_kat.json")))
    idl, sigma = ao.ideal_sigma()
    assert sigma == float(kat["weno_sigma"])
    assert list(idl) == [float(x) for x in kat["weno_idl"]]
    r0 = kat["reconstruct"][0]
    assert ao.reconstruct(r0["stencil"], 0) == float(r0["ind0"]) and ao.reconstruct(r0["stencil"], 1) == float(r0["ind1"])
    s = np.sin(0.3 * np.arange(5) + 0.1)
    r1 = kat["reconstruct"][1]
    assert ao.reconstruct(s, 0) == float(r1["ind0"]) and ao.reconstruct(s, 1) == float(r1["ind1"])


@pytest.mark.parametrize("name", sorted(mg.CASES))
def test_oracle_reproduces_golden_case_bitwise(name):
    c = mg.CASES[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    inputs, out, hydro, ncyc = mg.run_oracle(c)
    for k in inputs:
        assert np.array_equal(inputs[k], g["in_" + k]), ("input generator changed", k)
    assert list(g["ncycles"]) == ncyc
    assert np.array_equal(hydro, g["hydro"])
    for k in out:
        assert np.array_equal(out[k], g["out_" + k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(mg.CASES))
def test_gpu_matches_golden_case(name):
    import torch
    from pam_amd import Dycore, PamCoupler
    c = mg.CASES[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    tr, consts, zi, xlen, ylen, _ = mg.build_case(c)
    coupler = PamCoupler("cuda:0")
    coupler.set_option("crm_dt", c["crm_dt"])
    for k, v in consts.items():
        coupler.set_option(k, v)
    coupler.allocate_coupler_state(c["nz"], c["ny"], c["nx"], c["nens"])
    coupler.set_grid(xlen, ylen, zi)
    for n, p, m in tr:
        coupler.add_tracer(n, "", p, m)
    dycore = Dycore()
    dycore.init(coupler)
    coupler.load_fields({k[3:]: g[k] for k in g.files if k.startswith("in_")})   # inputs from the fixture itself
    if not c["mode_a"]:
        coupler.set_option("balance_hydrostasis_with_gravity", False)
    dycore.declare_current_profile_as_hydrostatic(coupler)
    hv = coupler.dm.get("variable_gravity" if c["mode_a"] else "hy_dens_cells", readonly=True).cpu().numpy()
    assert np.abs(hv - g["hydro"]).max() <= 1e-12 * np.abs(g["hydro"]).max()
    for n in g["ncycles"]:
        assert dycore.timeStep(coupler) == int(n)
    torch.cuda.synchronize()
    got = coupler.dump_fields()
    names = [t[0] for t in tr]
    exp = {k: g["out_" + k] for k in ("density_dry", "temp", "uvel", "vvel", "wvel")}
    exp["tracers"] = g["out_tracers"]
    compare(got, exp, names, int(np.sum(g["ncycles"])), "golden_" + name)      # tests/parity_gate.py: the measured-curve gate
    dycore.finalize(coupler)
