"""CPU-only, and only where the reference tree is mounted (this container; skipped on the GPU box): every literal of the
reference's generated matrices that the hot path uses (dynamics/awfl/TransformMatrices.h: sten_to_coefs<5,5> :970,
coefs_to_gll_lower<5,2> :1132, weno_lower_sten_to_coefs<3,3,3> :1218, coefs_to_tv<3> :188, coefs_to_tv<5> :871,
get_gll_points<9> :4113, get_gll_weights<9> :4126) against the constants this repository derives independently from
exact rationals (tools/gen_constants.py -> awfl_constants.h).  The reference file is only READ as text here; nothing of
it is stored in the repository.  This pins the constants of the oracle and of the HIP kernels to the reference itself."""
import os
import re

import numpy as np
import pytest

REF = "/root/reference/dynamics/awfl/TransformMatrices.h"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not mounted (GPU box)")


def _function_body(text, signature):
    i = text.index(signature)
    j = text.index("\n  }\n", i)
    return text[i:j]


def _matrix(text, signature, shape):
    body = _function_body(text, signature)
    a = np.full(shape, np.nan)
    for m in re.finditer(r"rslt\(([\d,]+)\)=(-?[\d.]+(?:e-?\d+)?)", body):
        a[tuple(int(x) for x in m.group(1).split(","))] = float(m.group(2))
    assert not np.isnan(a).any(), signature
    return a


def _ours(name, shape=None):
    hdr = open(os.path.join(ROOT, "pam_amd", "csrc", "awfl_constants.h")).read()
    assert hdr == open(os.path.join(ROOT, "oracle", "awfl_constants.h")).read()      # kernels and oracle share the values
    m = re.search(r"#define " + name + r" (.*?)\n(?=#define|/\*|\n#endif)", hdr, flags=re.S)
    txt = m.group(1).replace("\\\n", " ")
    vals = [float(v) for v in re.findall(r"-?\d+\.?\d*(?:e-?\d+)?", txt)]
    return np.array(vals).reshape(shape) if shape else vals[0]


def test_generated_constants_equal_the_reference_literals():
    ref = open(REF).read()
    s5 = _matrix(ref, "void sten_to_coefs(SArray<FP,2,5,5> &rslt)", (5, 5))
    assert np.abs(s5 - _ours("AWFL_STEN_TO_COEFS_INIT", (5, 5))).max() <= 1e-16
    c2g = _matrix(ref, "void coefs_to_gll_lower(SArray<FP,2,5,2> &rslt)", (5, 2))
    assert np.abs(c2g - _ours("AWFL_COEFS_TO_GLL_INIT", (5, 2))).max() <= 1e-16
    w3 = _matrix(ref, "void weno_lower_sten_to_coefs(SArray<FP,3,3,3,3> &rslt)", (3, 3, 3))
    assert np.abs(w3 - _ours("AWFL_WENO_LOWER_INIT", (3, 3, 3))).max() <= 1e-16
    pts = _matrix(ref, "void get_gll_points(SArray<FP,1,9> &rslt)", (9,))
    wts = _matrix(ref, "void get_gll_weights(SArray<FP,1,9> &rslt)", (9,))
    assert np.abs(pts - _ours("AWFL_GLL9_PTS_INIT", (9,))).max() <= 1e-16
    assert np.abs(wts - _ours("AWFL_GLL9_WTS_INIT", (9,))).max() <= 1e-16
    # total-variation quadratic forms: coefficient of each monomial in the reference expression
    tv3 = _function_body(ref, "FP coefs_to_tv(SArray<FP,1,3> &a)")
    tv5 = _function_body(ref, "FP coefs_to_tv(SArray<FP,1,5> &a)")

    def coef(body, i, j):
        m = re.search(r"(-?[\d.]+)_fp\*\(?a\(%d\)\*a\(%d\)\)?" % (i, j), body)
        return float(m.group(1)) if m else 0.0
    assert coef(tv3, 1, 1) == 1.0 and abs(coef(tv3, 2, 2) - _ours("AWFL_TV3_A2A2")) <= 1e-15
    for (i, j), name in {(1, 1): "AWFL_TV5_A1A1", (2, 2): "AWFL_TV5_A2A2", (1, 3): "AWFL_TV5_A1A3", (3, 3): "AWFL_TV5_A3A3",
                         (2, 4): "AWFL_TV5_A2A4", (4, 4): "AWFL_TV5_A4A4"}.items():
        assert abs(coef(tv5, i, j) - _ours(name)) <= 1e-13 * max(1.0, abs(_ours(name))), (i, j)
    # no other monomial appears in the reference's quartic form
    found = set(re.findall(r"a\((\d)\)\*a\((\d)\)", tv5))
    assert found == {("1", "1"), ("2", "2"), ("1", "3"), ("3", "3"), ("2", "4"), ("4", "4")}


def test_weno_ideal_weights_and_scalar_constants_equal_the_reference():
    """wenoSetIdealSigma<5> (WenoLimiter.h:37-43), the acoustic speed cs = 350 (Dycore.h:335), hs = (ord+1)/2 (Dycore.h:23)."""
    wl = open("/root/reference/dynamics/awfl/WenoLimiter.h").read()
    blk = wl[wl.index("} else if (ord == 5) {"):wl.index("} else if (ord == 7) {")]
    sigma = float(re.search(r"sigma = ([\d.]+)_fp", blk).group(1))
    idl = [float(x) for x in re.findall(r"idl\(\d\) = ([\d.]+)_fp", blk)]
    assert sigma == _ours("AWFL_WENO_SIGMA")
    assert idl == list(_ours("AWFL_WENO_IDL_INIT", (4,)))
    dy = open("/root/reference/dynamics/awfl/Dycore.h").read()
    assert re.search(r"real constexpr cs = 350;", dy)
    dev = open(os.path.join(ROOT, "pam_amd", "csrc", "awfl_device.h")).read()
    assert "const double cs = 350.0" in dev and "constexpr int HS = 3;" in dev
    assert re.search(r"static constexpr hs\s*=\s*\(ord\+1\)/2;", dy)      # ord = 5 -> 3 halo / ghost cells


KESSLER_LITERALS = ["36.34", "0.1364", "3.8", "17.27", "2.2", "0.875", "4093.", "1.6", "124.9", "0.2046", "0.525",
                    "2550000.", "540000.", "273.", "36.", "0.001", "0.8", "1.e-10"]


def test_kessler_and_module_literals_appear_in_reference_oracle_and_kernels():
    """Numeric literals of the Kessler scheme (physics/micro/kessler/Microphysics.h:346-457) and of the sponge layer
    defaults: each must be present in the reference text, in the oracle restatement and in the HIP kernels (a typo guard;
    the arithmetic itself is covered by the oracle-vs-HIP parity tests)."""
    ref = open("/root/reference/physics/micro/kessler/Microphysics.h").read()
    ora = open(os.path.join(ROOT, "oracle", "awfl_oracle.c")).read()
    hip = open(os.path.join(ROOT, "pam_amd", "csrc", "modules_kernels.hip")).read()
    for lit in KESSLER_LITERALS:
        pat = re.escape(lit.rstrip(".")) + r"(?![\d])"
        assert re.search(pat, ref), ("reference", lit)
        assert re.search(pat, ora), ("oracle", lit)
        assert re.search(pat, hip), ("kernels", lit)
    for name in ("R_d", "cp_d", "cp_v", "R_v", "p0", "grav"):      # the scheme's constants, Microphysics.h:66-71
        m = re.search(name + r"\s*=\s*([\d.e+]+)\s*;", ref)
        from pam_amd.micro import Microphysics
        assert float(m.group(1)) == getattr(Microphysics, name), name
    sp = open("/root/reference/pam_core/modules/sponge_layer.h").read()
    assert re.search(r"num_layers\s*=\s*5;", sp) and re.search(r"time_scale\s*=\s*60", sp)
