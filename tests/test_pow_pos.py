"""pow_pos_fast (pam_amd/csrc/awfl_device.h): x**y for positive x as every kernel of the step computes it (pressure, potential
temperature, hydrostatic ghosts: Dycore.h:310-321, :682-709, :1313-1387).  CPU: the function compiled for the host (same IEEE
operations: fma, frexp, ldexp, rint) against 80-bit powl.  GPU: the device evaluates the same bits as the host."""
import numpy as np
import pytest

import emu_harness as eh

EXPONENTS = [1003.0 / 716.0, 716.0 / 1003.0, 1003.0 / 716.0 - 1.0, 1.0 / (1003.0 / 716.0 - 1.0), 1004.64 / 717.598, 1.0, 0.5, 3.5]


def _samples(n, seed):
    rng = np.random.default_rng(seed)
    return np.exp2(rng.uniform(-9.0, 14.0, n))          # 2e-3 .. 1.6e4: densities, rho*theta, p/C0 and their powers


def _ulp_err(got, x, y):
    ref = np.power(x.astype(np.longdouble), np.longdouble(y))
    ulp = (np.nextafter(ref.astype(np.float64), np.inf) - ref.astype(np.float64)).astype(np.longdouble)
    return float(np.max(np.abs((got.astype(np.longdouble) - ref) / ulp)))


@pytest.mark.skipif(np.finfo(np.longdouble).nmant < 63, reason="needs 80-bit long double as the reference")
def test_pow_pos_fast_is_within_0p55_ulp_of_powl():
    worst = 0.0
    for i, y in enumerate(EXPONENTS):
        x = _samples(200000, 100 + i)
        worst = max(worst, _ulp_err(eh.emu_pow(x, y), x, y))
    print("pow_pos_fast: worst error %.4f ulp over %d samples" % (worst, 200000 * len(EXPONENTS)))
    assert worst <= 0.55


def test_pow_pos_fast_edge_values():
    y = 1003.0 / 716.0
    x = np.array([1.0, 2.0, 0.5, 1.0 - 2.0 ** -53, 1.0 + 2.0 ** -52, 4.0, 1e-300, 1e300, 0.0, -1.0, np.nan, np.inf, 5e-324])
    got = eh.emu_pow(x, y)
    assert np.isinf(got[11]) and got[11] > 0 and got[12] == 0.0    # +inf stays +inf; the smallest subnormal underflows to 0
    neg = eh.emu_pow(np.array([0.0, np.inf, 4.0]), -0.5)
    assert np.isinf(neg[0]) and neg[1] == 0.0 and abs(neg[2] - 0.5) <= 1e-16
    with np.errstate(invalid="ignore", over="ignore"):
        exp = np.power(x, y)
    assert got[0] == 1.0 and got[8] == 0.0 and np.isnan(got[9]) and np.isnan(got[10])
    assert got[6] == 0.0 and np.isinf(got[7])                    # under- and overflow like the C library
    assert np.all(np.abs(got[:6] - exp[:6]) <= 1.0 * np.spacing(exp[:6]))


@pytest.mark.gpu
def test_device_pow_equals_host_pow_bit_for_bit():
    import torch
    from pam_amd import idealized as idz
    from test_gpu_parity import _setup
    coupler, dycore, oracle, fo, names = _setup(2, 6, 1, 8, idz.TRACERS_NONE, idz.uniform_interfaces(8, 8000.0))
    for i, y in enumerate(EXPONENTS[:5]):
        x = _samples(100000, 7 + i)
        got = dycore.debug_pow(torch.from_numpy(x).to("cuda:0"), y)
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy(), eh.emu_pow(x, y)), y
    dycore.finalize(coupler)
