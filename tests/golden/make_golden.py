#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/.

Two kinds of fixture, kept apart because their provenance differs:

  survey_kat.json      REFERENCE-ARITHMETIC values recorded in SURVEY.md Appendix B (obtained from the reference's own
                       headers during the survey).  Copied here verbatim as data; this script only re-emits them.
  case_*.npz           ORACLE-GENERATED vectors: seeded inputs (built by pam_amd.idealized) and the coupler fields after
                       N Dycore::timeStep calls computed by oracle/awfl_oracle.c (gcc -O2 -ffp-contract=off).  They are
                       NOT reference outputs -- the reference cannot be built in this image (YAKL absent) -- they pin
                       the oracle against silent change between rounds and give the GPU tests a committed target.

Usage:  python tests/golden/make_golden.py      (rewrites the fixtures; run only when the oracle changes on purpose)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import awfl_oracle as ao          # noqa: E402
from pam_amd import idealized as idz          # noqa: E402

SURVEY_KAT = {
    "source": "SURVEY.md Appendix B (reference headers, serial run during the survey)",
    "weno_sigma": "0.73564225445964004",
    "weno_idl": ["0.00060224368852556992", "0.044303590476103952", "0.00060224368852556992", "0.95449192214684486"],
    "reconstruct": [
        {"stencil": [0, 0, 0.85, 1, 1], "ind0": "0.58122938412882086", "ind1": "1.0195943293936724"},
        {"stencil": "sin(0.3*i+0.1), i=0..4", "ind0": "0.52468366660293864", "ind1": "0.75407502647782132"},
    ],
    "bubble_32x1x60_nens2_5steps_max_abs_w": {"mode_A": "0.64954263988863503", "mode_B": "0.64953962145989119"},
    "resting_column_variable_gravity": 9.81,
}

CASES = {
    # name: nens, nx, ny, nz, tracer set, grid, mode_a, nsteps, crm_dt, constants, per-member dz
    "case_2d_nt1_uniform_A": dict(nens=3, nx=8, ny=1, nz=10, tr="none", grid=("uniform", 10000.0), mode_a=True,
                                  nsteps=2, crm_dt=2.0, consts="default", per_ens=False),
    "case_3d_nt4_stretched_B": dict(nens=2, nx=6, ny=6, nz=8, tr="kessler_shoc", grid=("stretched", 12000.0),
                                    mode_a=False, nsteps=2, crm_dt=2.0, consts="default", per_ens=False),
    "case_3d_nt10_perens_A_p3": dict(nens=3, nx=6, ny=4, nz=8, tr="p3_shoc", grid=("stretched", 12000.0), mode_a=True,
                                     nsteps=2, crm_dt=2.0, consts="p3", per_ens=True),
}
TRACERS = {"none": idz.TRACERS_NONE, "kessler_shoc": idz.TRACERS_KESSLER_SHOC, "p3_shoc": idz.TRACERS_P3_SHOC}
CONSTS = {"default": idz.CONSTS_DEFAULT, "p3": idz.CONSTS_P3}


def build_case(c):
    """Inputs of a case (deterministic: analytic profiles + splitmix64)."""
    tr, consts = TRACERS[c["tr"]], CONSTS[c["consts"]]
    nens, nx, ny, nz = c["nens"], c["nx"], c["ny"], c["nz"]
    zint = idz.uniform_interfaces(nz, c["grid"][1]) if c["grid"][0] == "uniform" else idz.stretched_interfaces(nz, c["grid"][1])
    xlen = nx * 500.0
    ylen = ny * 500.0 if ny > 1 else xlen
    f = idz.supercell_fields(nens, nx, ny, nz, zint, consts=consts, tracers=tr, magnitude=0.5)
    idz.add_tracer_blobs(f, tr, xlen, ylen, zint)
    zi = np.asarray(zint)[:, None] * np.ones((1, nens))
    if c["per_ens"]:
        zi = zi * (1 + 0.01 * np.arange(nens))[None, :]
    return tr, consts, zi, xlen, ylen, f


def run_oracle(c):
    tr, consts, zi, xlen, ylen, f = build_case(c)
    names, pos, mass, idwv = idz.tracer_flags(tr)
    o = ao.OracleDycore(c["nens"], c["nx"], c["ny"], c["nz"], xlen, ylen, np.diff(zi, axis=0), pos, mass, idwv, consts=consts)
    o.set_grav_balance(c["mode_a"])
    inputs = {k: v.copy() for k, v in f.items()}
    o.declare_current_profile_as_hydrostatic(f)
    hydro = (o.variable_gravity if c["mode_a"] else o.hy_dens_cells).copy()
    ncyc = []
    for _ in range(c["nsteps"]):
        n, _ = o.time_step(f, c["crm_dt"])
        ncyc.append(n)
    return inputs, f, hydro, ncyc


def main():
    with open(os.path.join(HERE, "survey_kat.json"), "w") as fh:
        json.dump(SURVEY_KAT, fh, indent=1)
    for name, c in CASES.items():
        inputs, out, hydro, ncyc = run_oracle(c)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), ncycles=np.array(ncyc), hydro=hydro,
                            **{"in_" + k: v for k, v in inputs.items()}, **{"out_" + k: v for k, v in out.items()})
        print(name, "ncycles", ncyc, "bytes", os.path.getsize(os.path.join(HERE, name + ".npz")))


if __name__ == "__main__":
    main()
