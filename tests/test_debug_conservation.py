"""The reference's runtime self-check (Dycore.h:36-58 compute_mass; under PAM_DEBUG :136-138 before, :224-251 after the sub-steps of a
timeStep: a WARNING when the mass of a variable of a member changed by more than 1e-10 relative and absolute) as the opt-in of the C ABI
(pam_amd_awfl_set_debug_conservation): silent on a clean run, fires for exactly the (variable, member) whose mass was tampered with."""
import numpy as np
import pytest

from pam_amd import idealized as idz

pytestmark = pytest.mark.gpu


def _setup(nens, nx, ny, nz, tr):
    from pam_amd import Dycore, PamCoupler
    zint = idz.stretched_interfaces(nz, 12000.0)
    xlen, ylen = nx * 500.0, (ny if ny > 1 else nx) * 500.0
    f = idz.supercell_fields(nens, nx, ny, nz, zint, tracers=tr, magnitude=0.5)
    idz.add_tracer_blobs(f, tr, xlen, ylen, zint)
    coupler = PamCoupler("cuda:0")
    coupler.set_option("crm_dt", 2.0)
    coupler.allocate_coupler_state(nz, ny, nx, nens)
    coupler.set_grid(xlen, ylen, zint)
    for n, p, m in tr:
        coupler.add_tracer(n, "", p, m)
    dycore = Dycore()
    dycore.init(coupler)
    coupler.load_fields(f)
    dycore.declare_current_profile_as_hydrostatic(coupler)
    return coupler, dycore


@pytest.mark.parametrize("nens,ny,fused", [(130, 4, True), (3, 1, True), (70, 3, False)], ids=["two_member_ranges", "small_2d", "three_kernel_stage"])
def test_conservation_check_is_silent_on_a_clean_run_and_fires_where_mass_was_changed(nens, ny, fused):
    from pam_amd import PamAmdError
    tr = idz.TRACERS_NONE            # water vapour alone: a smooth positive field -- every variable is conserved to round-off
    nt = len(tr)
    coupler, dycore = _setup(nens, 6, ny, 8, tr)
    dycore.set_fused_stage(fused)
    with pytest.raises(PamAmdError):
        dycore.conservation()                       # off by default: nothing to report
    dycore.timeStep(coupler)
    dycore.set_debug_conservation(True)
    for _ in range(3):
        dycore.timeStep(coupler)
        n, max_rel, _, _, report = dycore.conservation()
        assert n == 0 and report == "" and max_rel <= 1.0e-10, (n, max_rel, report)
    # rho of one cell of member m scaled by 1 + 1e-5 between the last stage and the final masses (rho*theta kept)
    m = nens - 2
    dycore.debug_inject_mass_fault(nt, 2, ny - 1, 3, m, 1.0 + 1.0e-5)
    dycore.timeStep(coupler)
    n, max_rel, wv, wm, report = dycore.conservation()
    assert n == 1 and (wv, wm) == (nt, m), (n, wv, wm, report)
    assert 1.0e-10 < max_rel < 1.0e-5
    assert report.startswith("WARNING: conservation violated variable,ensemble,rel_diff,mass_diff,init,final: %d , %d ," % (nt, m))
    # the tracer (water_vapor, variable 0) of member 0: variable 0 alone
    dycore.debug_inject_mass_fault(0, 3, 0, 2, 0, 1.001)
    dycore.timeStep(coupler)
    n, _, wv, wm, report = dycore.conservation()
    assert n == 1 and (wv, wm) == (0, 0), report
    # one-shot: the next step is clean again (the poked state is a legitimate new state)
    dycore.timeStep(coupler)
    assert dycore.conservation()[0] == 0
    dycore.set_debug_conservation(False)
    with pytest.raises(PamAmdError):
        dycore.conservation()
    dycore.finalize(coupler)


def test_blob_tracers_are_reported_as_the_reference_would_report_them():
    """Sharp-edged positive tracers (cloud, rain, tke blobs with exact zeros around them) gain a little mass per timeStep BY DESIGN of
    the reference: the max(0, .) clipping of the stage combines (Dycore.h:169-171) and the periodic-seam min() (:574-579, SURVEY quirk
    Q4).  The oracle does the same (tests/test_oracle_kat.py::test_mass_conservation_per_step bounds it at 1e-4), and so the reference's
    PAM_DEBUG build prints its WARNING for them; the check here reports exactly those variables and never rho, rho*theta or vapour."""
    tr = idz.TRACERS_KESSLER_SHOC
    nt = len(tr)
    coupler, dycore = _setup(3, 6, 1, 8, tr)
    dycore.set_debug_conservation(True)
    dycore.timeStep(coupler)
    n, max_rel, wv, _, report = dycore.conservation()
    assert 0 < n <= 3 * 3 and max_rel < 1.0e-3, (n, max_rel)
    lines = [l for l in report.split("\n") if l]
    assert len(lines) == n
    for l in lines:
        var = int(l.split(":")[2].split(",")[0])
        assert 1 <= var <= 3, l               # cloud_liquid, precip_liquid, tke; never 0 (vapour), nt (rho), nt + 1 (rho*theta)
    assert 1 <= wv <= 3
    dycore.finalize(coupler)
