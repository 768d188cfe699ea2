#!/usr/bin/env python3
"""Run-to-run determinism of the reference's CI configuration (65 x 1 x 50, nens = 1, dycore -> sponge -> Kessler) through the Python
host side, with every scheduling knob reachable: runs the same case R times, hashes every coupler field after every module call and
reports the first (step, module) at which a run leaves the first run's sequence.

  python tools/repro_ci_run.py [--runs R] [--steps S] [--nens N] [--xexchange lds|shuffle] [--tilefusion separate|inside|beside]
                               [--stateparts one|parts] [--ftileparts behind|beside] [--lanes member|flat] [--xkernels sweep|tile]
                               [--no-sponge] [--no-micro] [--fused 0|1]
"""
import argparse
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=8)
    ap.add_argument("--steps", type=int, default=90)
    ap.add_argument("--nens", type=int, default=1)
    ap.add_argument("--nx", type=int, default=65)
    ap.add_argument("--xexchange", default="auto")
    ap.add_argument("--tilefusion", default="auto")
    ap.add_argument("--stateparts", default="auto")
    ap.add_argument("--ftileparts", default="auto")
    ap.add_argument("--ftile", default="auto")
    ap.add_argument("--lanes", default="auto")
    ap.add_argument("--xkernels", default="auto")
    ap.add_argument("--fused", type=int, default=-1)
    ap.add_argument("--no-sponge", action="store_true")
    ap.add_argument("--no-micro", action="store_true")
    ap.add_argument("--sync", action="store_true", help="device-wide synchronisation after every module call")
    ap.add_argument("--final-only", action="store_true", help="hash the state once, at the end: nothing synchronises between the modules")
    ap.add_argument("--check", action="store_true", help="the dycore's conservation check on")
    a = ap.parse_args()
    import numpy as np
    import torch
    from pam_amd import Dycore, PamCoupler, Microphysics, modules
    dev = "cuda:0"
    nx, ny, nz, nens = a.nx, 1, 50, a.nens
    zint = np.linspace(0.0, 20000.0, nz + 1)

    def one_run():
        c = PamCoupler(dev)
        c.set_option("crm_dt", 20.0)
        c.set_option("gcm_physics_dt", 900.0)
        c.allocate_coupler_state(nz, ny, nx, nens)
        c.set_grid(128000.0, 64000.0, zint)
        micro = Microphysics()
        micro.init(c)
        d = Dycore()
        d.init(c)
        if a.lanes != "auto" or a.xkernels != "auto":
            d.set_lane_mapping(a.lanes, a.xkernels)
        if a.xexchange != "auto":
            d.set_x_exchange(a.xexchange)
        if a.tilefusion != "auto":
            d.set_tile_fusion(a.tilefusion)
        if a.stateparts != "auto":
            d.set_tile_state_parts(a.stateparts)
        if a.ftileparts != "auto":
            d.set_flux_tile_parts(a.ftileparts)
        if a.ftile != "auto":
            d.set_flux_tile(a.ftile)
        if a.fused >= 0:
            d.set_fused_stage(a.fused)
        dm = c.get_data_manager_device_readwrite()
        cols = modules.supercell_init(torch.from_numpy(zint).to(dev), c.get_option("R_d"), c.get_option("R_v"), c.get_option("grav"))
        for name, col in zip(("gcm_density_dry", "gcm_uvel", "gcm_vvel", "gcm_wvel", "gcm_temp", "gcm_water_vapor"), cols):
            dm.get(name).copy_(col[:, None].expand(nz, nens))
        modules.broadcast_initial_gcm_column(c)
        keep = modules.perturb_temperature(c, np.zeros(nens, dtype=np.int32), 0.1)
        names = ["density_dry", "uvel", "vvel", "wvel", "temp"] + c.get_tracer_names()

        def digest():
            torch.cuda.synchronize()
            h = hashlib.sha1()
            for n in names:
                h.update(dm.get(n).cpu().numpy().tobytes())
            return h.hexdigest()[:12]
        seq = [("init", digest())]
        if a.check:
            d.set_debug_conservation(True)
        nsub, worst = 0, 0.0
        for s in range(a.steps):
            if s % 45 == 0:
                d.declare_current_profile_as_hydrostatic(c)
            n = d.timeStep(c)
            nsub += n
            if a.check:
                worst = max(worst, d.conservation()[1])
            if a.sync:
                torch.cuda.synchronize()
            if not a.final_only:
                seq.append(("step %d dycore (%d sub-steps)" % (s, n), digest()))
            if not a.no_sponge:
                modules.sponge_layer(c)
                if not a.final_only:
                    seq.append(("step %d sponge" % s, digest()))
            if not a.no_micro:
                micro.timeStep(c)
                if not a.final_only:
                    seq.append(("step %d micro" % s, digest()))
        seq.append(("final, %d sub-steps, worst mass change %.2e" % (nsub, worst), digest()))
        mapping = d.get_lane_mapping()
        d.finalize(c)
        del keep
        return seq, mapping
    ref, mapping = one_run()
    print("lane mapping:", mapping)
    bad = 0
    for r in range(1, a.runs):
        seq, _ = one_run()
        first = next((i for i, (x, y) in enumerate(zip(ref, seq)) if x != y), None)
        if first is None:
            print("run %d: identical (%d checkpoints)" % (r, len(seq)))
        else:
            bad += 1
            print("run %d: FIRST DIFFERENCE at checkpoint %d: %s  (ref %s)" % (r, first, seq[first], ref[first]))
    print("diverging runs: %d of %d" % (bad, a.runs - 1))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
