"""The C++ host side above the C ABI (pam_amd/csrc/host: minimal pam::PamCoupler work-alike + the plug-in class
dynamics/awfl_amd/Dycore.h) driven by examples/driver.cpp, which mirrors the reference driver's call sequence
(standalone/mmf_simplified/driver.cpp:120-191,237-272).  The binary is built by __graft_entry__.build()."""
import os
import struct
import subprocess

import numpy as np
import pytest

from pam_amd import idealized as idz
from parity_gate import compare

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "examples", "driver")


@pytest.mark.gpu
@pytest.mark.parametrize("mode_a", [True, False])
def test_cpp_driver_matches_oracle(tmp_path, mode_a):
    from oracle import awfl_oracle as ao
    assert os.path.exists(DRIVER), "examples/driver missing: run __graft_entry__.build()"
    nens, nx, ny, nz, nsteps, crm_dt = 5, 7, 4, 9, 2, 2.0
    tr = idz.TRACERS_KESSLER_SHOC
    names, pos, mass, idwv = idz.tracer_flags(tr)
    zint = idz.stretched_interfaces(nz, 12000.0)
    xlen, ylen = nx * 500.0, ny * 500.0
    f = idz.supercell_fields(nens, nx, ny, nz, zint, tracers=tr, magnitude=0.5)
    idz.add_tracer_blobs(f, tr, xlen, ylen, zint)
    c = idz.CONSTS_DEFAULT
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(inp, "wb") as fh:
        fh.write(struct.pack("<8q", nens, nx, ny, nz, len(tr), nsteps, int(mode_a), 1))
        fh.write(struct.pack("<3d", xlen, ylen, crm_dt))
        fh.write(struct.pack("<6d", c["R_d"], c["cp_d"], c["R_v"], c["cp_v"], c["p0"], c["grav"]))
        fh.write(np.asarray(zint, dtype="<f8").tobytes())
        fh.write(bytes(bytearray(v for t in range(len(tr)) for v in (int(pos[t]), int(mass[t])))))
        fh.write(struct.pack("<q", idwv))
        for k in ("density_dry", "uvel", "vvel", "wvel", "temp"):
            fh.write(f[k].astype("<f8").tobytes())
        for t in range(len(tr)):
            fh.write(f["tracers"][t].astype("<f8").tobytes())
    r = subprocess.run([DRIVER, inp, outp], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "SSPRK3+WENO+FV A-grid" in r.stdout
    raw = np.fromfile(outp, dtype="<f8").reshape(5 + len(tr), nz, ny, nx, nens)
    o = ao.OracleDycore(nens, nx, ny, nz, xlen, ylen, np.diff(zint), pos, mass, idwv)
    o.set_grav_balance(mode_a)
    o.declare_current_profile_as_hydrostatic(f)
    nsub = 0
    for _ in range(nsteps):
        nsub += o.time_step(f, crm_dt)[0]
    got = {"density_dry": raw[0], "uvel": raw[1], "vvel": raw[2], "wvel": raw[3], "temp": raw[4], "tracers": raw[5:]}
    compare(got, f, [t[0] for t in tr], nsub)           # tests/parity_gate.py: the measured-curve gate


@pytest.mark.gpu
def test_cpp_dycore_converts_with_the_reference_argument_lists(tmp_path):
    """convert_coupler_to_dynamics(coupler, state, tracers) / convert_dynamics_to_coupler(coupler, state, tracers) -- the
    signatures E3SM's pam_driver calls (awfl/Dycore.h:1336-1338, :1281-1283) -- through the C++ plug-in class: coupler -> halo'd
    arrays, coupler fields overwritten with NaN patterns, arrays -> coupler; no time step.  The round trip reproduces the inputs
    (tracers bit for bit, the rest to the rounding of the two pow calls)."""
    assert os.path.exists(DRIVER), "examples/driver missing: run __graft_entry__.build()"
    nens, nx, ny, nz = 5, 7, 4, 9
    tr = idz.TRACERS_KESSLER_SHOC
    names, pos, mass, idwv = idz.tracer_flags(tr)
    zint = idz.stretched_interfaces(nz, 12000.0)
    xlen, ylen = nx * 500.0, ny * 500.0
    f = idz.supercell_fields(nens, nx, ny, nz, zint, tracers=tr, magnitude=0.5)
    idz.add_tracer_blobs(f, tr, xlen, ylen, zint)
    c = idz.CONSTS_DEFAULT
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(inp, "wb") as fh:
        fh.write(struct.pack("<8q", nens, nx, ny, nz, len(tr), 0, 1 | 8, 1))      # no steps; bit 8: the halo-array round trip
        fh.write(struct.pack("<3d", xlen, ylen, 2.0))
        fh.write(struct.pack("<6d", c["R_d"], c["cp_d"], c["R_v"], c["cp_v"], c["p0"], c["grav"]))
        fh.write(np.asarray(zint, dtype="<f8").tobytes())
        fh.write(bytes(bytearray(v for t in range(len(tr)) for v in (int(pos[t]), int(mass[t])))))
        fh.write(struct.pack("<q", idwv))
        for k in ("density_dry", "uvel", "vvel", "wvel", "temp"):
            fh.write(f[k].astype("<f8").tobytes())
        for t in range(len(tr)):
            fh.write(f["tracers"][t].astype("<f8").tobytes())
    r = subprocess.run([DRIVER, inp, outp], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(outp, dtype="<f8").reshape(5 + len(tr), nz, ny, nx, nens)
    exp = [f["density_dry"], f["uvel"], f["vvel"], f["wvel"], f["temp"]] + [f["tracers"][t] for t in range(len(tr))]
    for i, e in enumerate(exp):
        assert np.isfinite(raw[i]).all(), i
        if i >= 5:
            assert np.array_equal(raw[i], e), i
        else:
            assert np.abs(raw[i] - e).max() <= 1e-14 * max(np.abs(e).max(), 1e-300), (i, np.abs(raw[i] - e).max())


@pytest.mark.gpu
def test_cpp_driver_crm_loop_dycore_sponge_kessler(tmp_path):
    """The CRM step loop of the reference driver minus SGS (driver.cpp:248-253): dycore -> sponge_layer -> Kessler
    micro, all three through their C++ plug-in mirrors, against the same sequence of oracle calls."""
    from oracle import awfl_oracle as ao
    nens, nx, ny, nz, nsteps, crm_dt = 4, 8, 1, 20, 3, 4.0
    tr = (("water_vapor", True, True), ("cloud_liquid", True, True), ("precip_liquid", True, True))
    names, pos, mass, idwv = idz.tracer_flags(tr)
    consts = dict(R_d=287.0, cp_d=1003.0, R_v=461.0, cp_v=1859.0, p0=1.0e5, grav=9.81)    # Microphysics.h:66-71
    zint = idz.stretched_interfaces(nz, 15000.0)
    zi = np.ascontiguousarray(np.broadcast_to(zint[:, None], (nz + 1, nens)))
    zm = 0.5 * (zi[:-1] + zi[1:])
    xlen, ylen = nx * 500.0, ny * 500.0
    f = idz.supercell_fields(nens, nx, ny, nz, zint, tracers=tr, magnitude=0.5, consts=consts)
    f["tracers"][0] *= 1.0 + 0.5 * np.cos(np.arange(nx))[None, None, :, None] ** 2       # supersaturate some columns
    f["tracers"][2][0:8] = 2e-3 * f["density_dry"][0:8]                                    # rain that reaches the ground
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(inp, "wb") as fh:
        fh.write(struct.pack("<8q", nens, nx, ny, nz, 3, nsteps, 1 | 2 | 4, 0))
        fh.write(struct.pack("<3d", xlen, ylen, crm_dt))
        fh.write(struct.pack("<6d", *([0.0] * 6)))
        fh.write(np.asarray(zint, dtype="<f8").tobytes())
        fh.write(bytes(bytearray([1, 1] * 3)))
        fh.write(struct.pack("<q", idwv))
        for k in ("density_dry", "uvel", "vvel", "wvel", "temp"):
            fh.write(f[k].astype("<f8").tobytes())
        for t in range(3):
            fh.write(f["tracers"][t].astype("<f8").tobytes())
    r = subprocess.run([DRIVER, inp, outp], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(outp, dtype="<f8")
    ncell = nz * ny * nx * nens
    got = raw[:8 * ncell].reshape(8, nz, ny, nx, nens)
    got_precl = raw[8 * ncell:].reshape(ny, nx, nens)
    o = ao.OracleDycore(nens, nx, ny, nz, xlen, ylen, np.diff(zint), pos, mass, idwv, consts=consts)
    o.declare_current_profile_as_hydrostatic(f)
    for _ in range(nsteps):
        o.time_step(f, crm_dt)
        ao.sponge_layer(f, zi, zm, crm_dt)
        trc = [np.ascontiguousarray(f["tracers"][t]) for t in range(3)]
        precl, _ = ao.kessler(trc[0], trc[1], trc[2], f["density_dry"], f["temp"], zm, crm_dt, consts)
        for t in range(3):
            f["tracers"][t] = trc[t]
    exp = [f["density_dry"], f["uvel"], f["vvel"], f["wvel"], f["temp"]] + [f["tracers"][t] for t in range(3)]
    assert precl.max() > 0 and f["tracers"][1].max() > 0      # it rained and cloud formed
    for i, e in enumerate(exp):
        tol = 1e-11 if i in (0, 4, 5) else 1e-8
        assert np.abs(got[i] - e).max() <= tol * max(np.abs(e).max(), 1e-300), i
    assert np.abs(got_precl - precl).max() <= 1e-10 * precl.max()


@pytest.mark.gpu
def test_cpp_driver_spam_surface_swap_in(tmp_path):
    """BASELINE config C5 (boundary only): the SAME examples/driver.cpp, built against dynamics/spam_surface/Dycore.h
    (SPAM's member set: pre_time_loop, update_dt, non-const finalize; numerics out of scope) with -DPAMC_DYCORE, runs the
    CRM loop with the same modules and microphysics: the coupler surface is dycore-agnostic.  The stub advances nothing, so
    the result is exactly sponge_layer + Kessler applied to the input (oracle sequence without the dycore call)."""
    from oracle import awfl_oracle as ao
    drv = os.path.join(ROOT, "examples", "driver_spam")
    assert os.path.exists(drv), "examples/driver_spam missing: run __graft_entry__.build()"
    nens, nx, ny, nz, nsteps, crm_dt = 4, 8, 1, 20, 2, 4.0
    tr = (("water_vapor", True, True), ("cloud_liquid", True, True), ("precip_liquid", True, True))
    names, pos, mass, idwv = idz.tracer_flags(tr)
    consts = dict(R_d=287.0, cp_d=1003.0, R_v=461.0, cp_v=1859.0, p0=1.0e5, grav=9.81)
    zint = idz.stretched_interfaces(nz, 15000.0)
    zi = np.ascontiguousarray(np.broadcast_to(zint[:, None], (nz + 1, nens)))
    zm = 0.5 * (zi[:-1] + zi[1:])
    xlen, ylen = nx * 500.0, ny * 500.0
    f = idz.supercell_fields(nens, nx, ny, nz, zint, tracers=tr, magnitude=0.5, consts=consts)
    f["tracers"][0] *= 1.0 + 0.5 * np.cos(np.arange(nx))[None, None, :, None] ** 2
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(inp, "wb") as fh:
        fh.write(struct.pack("<8q", nens, nx, ny, nz, 3, nsteps, 1 | 2 | 4, 0))
        fh.write(struct.pack("<3d", xlen, ylen, crm_dt))
        fh.write(struct.pack("<6d", *([0.0] * 6)))
        fh.write(np.asarray(zint, dtype="<f8").tobytes())
        fh.write(bytes(bytearray([1, 1] * 3)))
        fh.write(struct.pack("<q", idwv))
        for k in ("density_dry", "uvel", "vvel", "wvel", "temp"):
            fh.write(f[k].astype("<f8").tobytes())
        for t in range(3):
            fh.write(f["tracers"][t].astype("<f8").tobytes())
    r = subprocess.run([drv, inp, outp], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Dycore: SPAM++" in r.stdout
    ncell = nz * ny * nx * nens
    got = np.fromfile(outp, dtype="<f8")[:8 * ncell].reshape(8, nz, ny, nx, nens)
    for _ in range(nsteps):
        ao.sponge_layer(f, zi, zm, crm_dt)
        trc = [np.ascontiguousarray(f["tracers"][t]) for t in range(3)]
        ao.kessler(trc[0], trc[1], trc[2], f["density_dry"], f["temp"], zm, crm_dt, consts)
        for t in range(3):
            f["tracers"][t] = trc[t]
    exp = [f["density_dry"], f["uvel"], f["vvel"], f["wvel"], f["temp"]] + [f["tracers"][t] for t in range(3)]
    for i, e in enumerate(exp):
        assert np.abs(got[i] - e).max() <= 1e-12 * max(np.abs(e).max(), 1e-300), i


def _write_input(path, f, tr, zint, xlen, ylen, crm_dt, nsteps, flags, consts=None):
    names, pos, mass, idwv = idz.tracer_flags(tr)
    nz, ny, nx, nens = f["density_dry"].shape
    c = consts or idz.CONSTS_DEFAULT
    with open(path, "wb") as fh:
        fh.write(struct.pack("<8q", nens, nx, ny, nz, len(tr), nsteps, flags, 1))
        fh.write(struct.pack("<3d", xlen, ylen, crm_dt))
        fh.write(struct.pack("<6d", c["R_d"], c["cp_d"], c["R_v"], c["cp_v"], c["p0"], c["grav"]))
        fh.write(np.asarray(zint, dtype="<f8").tobytes())
        fh.write(bytes(bytearray(v for t in range(len(tr)) for v in (int(pos[t]), int(mass[t])))))
        fh.write(struct.pack("<q", idwv))
        for k in ("density_dry", "uvel", "vvel", "wvel", "temp"):
            fh.write(f[k].astype("<f8").tobytes())
        for t in range(len(tr)):
            fh.write(f["tracers"][t].astype("<f8").tobytes())


@pytest.mark.gpu
@pytest.mark.parametrize("nens,ranks", [(130, (2, 3)), (6, (2, 4))], ids=["member_lane_shards", "flat_lane_shards"])
def test_cpp_driver_sharded_over_handles_equals_unsharded_bit_for_bit(tmp_path, nens, ranks):
    """examples/driver --gpus N: N host threads, one coupler + one dycore handle each (hipSetDevice before init; on this 1-GPU box
    they share device 0), the ensemble sharded by member index, the dynamics step = min over N host doubles -- no collective library
    (Dycore.h:86-101,141-145; standalone/mmf_simplified/driver.cpp:237-272).  One member is made the CFL-limiting
    one, so that a shard without it would sub-cycle differently if the exchange were missing.  Results equal the single-handle run
    bit for bit, with even and ragged shard sizes (130 -> 65+65 and 44+43+43; 6 -> 3+3 and 2+2+1+1)."""
    assert os.path.exists(DRIVER), "examples/driver missing: run __graft_entry__.build()"
    nx, ny, nz, nsteps, crm_dt = 8, 4, 10, 2, 2.0
    tr = idz.TRACERS_KESSLER_SHOC
    zint = idz.stretched_interfaces(nz, 12000.0)
    xlen, ylen = nx * 500.0, ny * 500.0
    f = idz.supercell_fields(nens, nx, ny, nz, zint, tracers=tr, magnitude=0.5)
    idz.add_tracer_blobs(f, tr, xlen, ylen, zint)
    f["uvel"][..., 1] += 40.0                      # member 1 alone sets the ensemble's CFL step
    inp = str(tmp_path / "in.bin")
    _write_input(inp, f, tr, zint, xlen, ylen, crm_dt, nsteps, 1)
    outs = []
    for n in (1,) + tuple(ranks):
        outp = str(tmp_path / ("out%d.bin" % n))
        r = subprocess.run([DRIVER, "--gpus", str(n), inp, outp], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        outs.append(np.fromfile(outp, dtype="<f8"))
    assert np.isfinite(outs[0]).all() and not np.array_equal(outs[0][:f["density_dry"].size], f["density_dry"].ravel())
    for o in outs[1:]:
        assert np.array_equal(outs[0], o)


@pytest.mark.gpu
def test_cpp_driver_bench_mode_tiles_the_input_and_reports_one_json_line(tmp_path):
    """--tile R --bench K W: what bench.py --launcher cpp runs"""
    import json
    nens, nx, ny, nz = 4, 8, 1, 10
    tr = idz.TRACERS_NONE
    zint = idz.stretched_interfaces(nz, 12000.0)
    f = idz.supercell_fields(nens, nx, ny, nz, zint, tracers=tr, magnitude=0.5)
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    _write_input(inp, f, tr, zint, nx * 500.0, nx * 500.0, 2.0, 0, 1)
    r = subprocess.run([DRIVER, "--gpus", "2", "--tile", "3", "--bench", "2", "1", inp, outp], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    d = json.loads(line[0])
    assert d["launcher"] == "cpp" and d["ranks"] == 2 and d["nens_total"] == 12 and d["steps"] == 2 and d["seconds"] > 0 and d["substeps"] >= 2
    assert np.fromfile(outp, dtype="<f8").size == 6 * nz * ny * nx * 12


def _kessler_case(nens=6):
    """a CRM ensemble in which ONE member carries heavy rain on a fine vertical grid: that member alone pushes the Kessler
    sedimentation sub-cycle count (rainsplit = ceil(dt / min over ALL columns of 0.8 dz / v_rain), Microphysics.h:385-390) above 1"""
    nx, ny, nz, crm_dt = 8, 1, 24, 8.0
    tr = (("water_vapor", True, True), ("cloud_liquid", True, True), ("precip_liquid", True, True))
    consts = dict(R_d=287.0, cp_d=1003.0, R_v=461.0, cp_v=1859.0, p0=1.0e5, grav=9.81)    # Microphysics.h:66-71
    zint = idz.uniform_interfaces(nz, 1200.0)
    f = idz.supercell_fields(nens, nx, ny, nz, zint, tracers=tr, magnitude=0.5, consts=consts)
    f["tracers"][0] *= 1.0 + 0.3 * np.cos(np.arange(nx))[None, None, :, None] ** 2
    f["tracers"][2][0:10, ..., nens - 2] = 4e-3 * f["density_dry"][0:10, ..., nens - 2]    # the last-but-one member rains hard
    f["tracers"][2][0:10, ..., 0] = 1e-7 * f["density_dry"][0:10, ..., 0]                  # member 0: drizzle
    return nx, ny, nz, crm_dt, tr, consts, zint, f


def test_kessler_case_of_the_sharded_driver_test_has_shard_dependent_rainsplit():
    """CPU (oracle): the input of the test below really separates the shards -- the whole ensemble needs more sedimentation
    sub-cycles than its first half would choose on its own"""
    import copy
    from oracle import awfl_oracle as ao
    nx, ny, nz, crm_dt, tr, consts, zint, f = _kessler_case()
    nens = f["temp"].shape[-1]
    zm = np.ascontiguousarray(np.broadcast_to((0.5 * (zint[:-1] + zint[1:]))[:, None], (nz, nens)))

    def split_of(lo, hi):
        g = copy.deepcopy(f)
        a = [np.ascontiguousarray(g["tracers"][t][..., lo:hi]) for t in range(3)]
        return ao.kessler(a[0], a[1], a[2], np.ascontiguousarray(g["density_dry"][..., lo:hi]), np.ascontiguousarray(g["temp"][..., lo:hi]),
                          np.ascontiguousarray(zm[:, lo:hi]), crm_dt, consts)[1]
    assert split_of(0, nens) >= 2 and split_of(0, nens // 2) == 1 and split_of(nens // 2, nens) == split_of(0, nens)


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3])
def test_cpp_driver_sharded_crm_loop_with_kessler_equals_unsharded_bit_for_bit(tmp_path, ranks):
    """`driver --gpus N` with the whole CRM loop (flags 1|2|4: dycore -> sponge_layer -> Kessler).  Kessler's
    sub-cycle count is a minimum over ALL members like the dycore's dt (Microphysics.h:385-390): the ranks exchange
    Microphysics::max_stable_dt through the same HostMin and pass ONE rainsplit to Microphysics::timeStep.  Without that exchange the
    shard without the raining member sub-cycles once instead of twice and its fields differ from the unsharded run."""
    assert os.path.exists(DRIVER), "examples/driver missing: run __graft_entry__.build()"
    nx, ny, nz, crm_dt, tr, consts, zint, f = _kessler_case()
    inp = str(tmp_path / "in.bin")
    _write_input(inp, f, tr, zint, nx * 500.0, nx * 500.0, crm_dt, 2, 1 | 2 | 4, consts=None)
    outs = []
    for n in (1, ranks):
        outp = str(tmp_path / ("out%d.bin" % n))
        r = subprocess.run([DRIVER, "--gpus", str(n), inp, outp], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        outs.append(np.fromfile(outp, dtype="<f8"))
    ncell = nz * ny * nx * f["temp"].shape[-1]
    assert np.isfinite(outs[0]).all() and outs[0][8 * ncell:].max() > 0            # it rained (precl is appended to the output)
    assert np.array_equal(outs[0], outs[1])
