"""The drop-in boundary: the C++ plug-in classes must compile inside PAM, i.e. call nothing on the coupler /
DataManager that PAM's own headers lack, and expose the reference `Dycore`'s signatures.

YAKL is absent, so the headers cannot be compiled against the reference's pam_core here; instead
  * every `coupler.X(` / `dm.X(` member used by the host-side C++ sources is looked up in the member lists of
    pam_core/pam_coupler.h and pam_core/DataManager.h.  The lists below are DATA extracted from those headers (names
    only); where the reference tree is mounted they are re-extracted live and must agree;
  * the same members must exist in the work-alike pam_amd/csrc/host/pam_coupler.h this repository compiles against;
  * the Dycore member signatures are compared with the reference's own lines (live) and with their recorded form;
  * INTEGRATION.md section 2 must be the shipped header, verbatim."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "pam_amd", "csrc", "host")
REF = "/root/reference"

# member names of pam::PamCoupler (pam_core/pam_coupler.h) and pam::DataManager (pam_core/DataManager.h)
REF_COUPLER = {"add_option", "add_tracer", "allocate_coupler_state", "delete_option", "finalize",
               "get_data_manager_device_readonly", "get_data_manager_device_readwrite", "get_data_manager_host_readonly",
               "get_data_manager_host_readwrite", "get_dx", "get_dy", "get_ncrms", "get_nens", "get_num_tracers", "get_nx",
               "get_ny", "get_nz", "get_option", "get_tracer_info", "get_tracer_names", "get_xlen", "get_ylen",
               "make_option_readonly", "option_exists", "run_module", "set_grid", "set_option", "tracer_exists"}
REF_DM = {"add_dimension", "clean_all_entries", "clean_entry", "entry_exists", "entry_is_dirty", "finalize", "get",
          "get_collapsed", "get_dimension_size", "get_dirty_entries", "get_lev_col", "get_shape", "is_read_only",
          "make_readonly", "register_and_allocate", "register_existing", "unregister_and_deallocate", "validate",
          "validate_all"}

SOURCES = [os.path.join(HOST, "dynamics", "awfl_amd", "Dycore.h"), os.path.join(HOST, "dynamics", "spam_surface", "Dycore.h"),
           os.path.join(HOST, "modules", "sponge_layer.h"), os.path.join(HOST, "modules", "gcm_forcing.h"),
           os.path.join(HOST, "modules", "broadcast_initial_gcm_column.h"), os.path.join(HOST, "modules", "perturb_temperature.h"),
           os.path.join(HOST, "physics", "micro", "kessler_amd", "Microphysics.h"), os.path.join(ROOT, "examples", "driver.cpp")]


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def _used_members(text):
    text = _strip_comments(text)
    aliases = set(re.findall(r"PamCoupler\s*(?:const\s*)?&\s*(\w+)", text)) | {"coupler"}   # every name a coupler is bound to
    coupler = set(re.findall(r"\b(?:%s)\s*\.\s*([A-Za-z_]\w*)\s*[<(]" % "|".join(sorted(aliases)), text))
    # chained: coupler.get_data_manager_device_readonly().get<...>
    dm = set(re.findall(r"\bdm\s*\.\s*([A-Za-z_]\w*)\s*[<(]", text))
    dm |= set(re.findall(r"get_data_manager_device_read\w+\(\)\s*\.\s*([A-Za-z_]\w*)\s*[<(]", text))
    return coupler, dm


def _declared(path):
    return set(re.findall(r"\b([A-Za-z_]\w*)\s*\(", _strip_comments(open(path).read())))


@pytest.mark.parametrize("src", SOURCES, ids=[os.path.relpath(s, ROOT) for s in SOURCES])
def test_host_sources_call_only_members_the_reference_has(src):
    coupler, dm = _used_members(open(src).read())
    assert coupler or dm, "scan found no coupler/DataManager calls: the regexes no longer match the sources"
    assert coupler <= REF_COUPLER, "not members of pam::PamCoupler: %s" % sorted(coupler - REF_COUPLER)
    assert dm <= REF_DM, "not members of pam::DataManager: %s" % sorted(dm - REF_DM)
    ours = _declared(os.path.join(HOST, "pam_coupler.h"))
    assert (coupler | dm) <= ours, "missing from the work-alike pam_coupler.h: %s" % sorted((coupler | dm) - ours)


def test_the_scan_catches_an_invented_member():
    coupler, dm = _used_members("auto &dm = coupler.get_data_manager_device_readwrite(); dm.unregister(name); coupler.frob<int>(1);")
    assert "unregister" in dm - REF_DM and "frob" in coupler - REF_COUPLER


def test_workalike_declares_nothing_under_an_invented_name():
    """every public member of the work-alike coupler/DataManager carries a reference name (round-1 shipped `unregister`)"""
    text = _strip_comments(open(os.path.join(HOST, "pam_coupler.h")).read())
    body = text[text.index("class DataManager"):text.index("class PamCoupler")]
    names = set(re.findall(r"^\s+(?:template\s*<[^>]*>\s*)?(?:[\w:<>&\*, ]+\s+)?([a-z_]\w*)\s*\([^;]*\)\s*(?:const\s*)?\{", body, flags=re.M))
    names -= {"if", "for", "endrun", "get"} | {"DataManager"}
    assert names and names <= REF_DM, sorted(names - REF_DM)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted (GPU box)")
def test_recorded_member_lists_match_the_reference_headers():
    pc = _declared(os.path.join(REF, "pam_core", "pam_coupler.h"))
    dm = _declared(os.path.join(REF, "pam_core", "DataManager.h"))
    assert REF_COUPLER <= pc, sorted(REF_COUPLER - pc)
    assert REF_DM <= dm, sorted(REF_DM - dm)
    for invented in ("unregister", "deallocate_entry"):
        assert invented not in REF_DM


# signature -> the reference line it must match (dynamics/awfl/Dycore.h), whitespace-insensitively
AWFL_SIGNATURES = {
    "void init(pam::PamCoupler &coupler, bool verbose = false)": 835,
    "void timeStep(pam::PamCoupler &coupler)": 107,
    "real compute_time_step(pam::PamCoupler const &coupler, real cfl = 0.8) const": 65,
    "void declare_current_profile_as_hydrostatic(pam::PamCoupler &coupler, bool use_gcm_data = false) const": 1392,
    "void convert_dynamics_to_coupler(pam::PamCoupler &coupler, realConst5d state, realConst5d tracers) const": 1281,
    "void convert_coupler_to_dynamics(pam::PamCoupler &coupler, real5d &state, real5d &tracers) const": 1336,
    "char const *dycore_name() const": 1544,
    "void finalize(pam::PamCoupler const &coupler) const": 1548,
}
SPAM_SIGNATURES = {   # dynamics/spam/Dycore.h (its `PamCoupler` is pam::PamCoupler through a using-declaration)
    "void init(PamCoupler &coupler, bool verbose = false)": 79,
    "void pre_time_loop(PamCoupler &coupler)": 169,
    "void update_dt(pam::PamCoupler &coupler)": 233,
    "real compute_time_step(PamCoupler const &coupler, real cfl_in = -1)": 244,
    "void timeStep(PamCoupler &coupler)": 248,
    "void finalize(PamCoupler &coupler)": 325,
    "const char *dycore_name() const": 327,
}


def _norm(s):
    return re.sub(r"\s+", "", s).replace("pam::", "")


def test_awfl_amd_dycore_has_the_reference_signatures():
    text = _norm(_strip_comments(open(os.path.join(HOST, "dynamics", "awfl_amd", "Dycore.h")).read()))
    for sig in AWFL_SIGNATURES:
        assert _norm(sig) + "{" in text, sig


def test_spam_surface_dycore_has_spams_signatures():
    text = _norm(_strip_comments(open(os.path.join(HOST, "dynamics", "spam_surface", "Dycore.h")).read()))
    for sig in SPAM_SIGNATURES:
        assert _norm(sig) + "{" in text, sig


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted (GPU box)")
@pytest.mark.parametrize("which", ["awfl", "spam"])
def test_recorded_signatures_are_the_reference_lines(which):
    sigs = AWFL_SIGNATURES if which == "awfl" else SPAM_SIGNATURES
    lines = open(os.path.join(REF, "dynamics", which, "Dycore.h")).read().split("\n")
    for sig, ln in sigs.items():
        got = _norm(" ".join(lines[ln - 1:ln + 2]))      # a signature may wrap
        assert got.startswith(_norm(sig)), (sig, lines[ln - 1])


def test_integration_md_lists_the_shipped_header():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sync_integration.py"), "--check"])
    assert r.returncode == 0, "INTEGRATION.md section 2 is stale: run python tools/sync_integration.py"
