"""The north-star loop (mc.LensedSimsMonteCarlo) on one GPU: simulations/s, per-stage HIP-event split and,
with --profile, the host-side cProfile of a few simulations.
    python3 tools/lensloop_bench.py [--side 4096] [--prec f32] [--nsims 8] [--profile]"""
import argparse
import cProfile
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from orphics_amd import cosmology, lensing, maps, mc          # noqa: E402
from orphics_amd.geometry import FlatGeometry                  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--side", type=int, default=4096)
ap.add_argument("--res", type=float, default=0.5)
ap.add_argument("--prec", default="f32")
ap.add_argument("--nsims", type=int, default=8)
ap.add_argument("--estimators", default="TT,EB")
ap.add_argument("--profile", action="store_true")
a = ap.parse_args()
shape = (3, a.side, a.side)
geom = FlatGeometry.from_res(shape, a.res)
theory = cosmology.default_theory()
t0 = time.perf_counter()
sims = lensing.FlatLensingSims(shape, geom, theory, 1.5, 1.0, pol=True, dtype=a.prec)
keep = {k: maps.mask_kspace(shape, geom, lmin=lo, lmax=hi) for k, (lo, hi) in (("T", (300., 2000.)), ("K", (20., 3500.)))}
q = lensing.qest(shape, geom, theory, noise2d=sims.ps_noise[0, 0], beam2d=sims.kbeam, kmask=keep["T"], noise2d_P=sims.ps_noise[1, 1],
                 kmask_P=keep["T"], kmask_K=keep["K"], pol=True, unlensed_equals_lensed=True, dtype=a.prec)
drv = mc.LensedSimsMonteCarlo(sims, q, np.linspace(20, 3500, 20), estimators=tuple(a.estimators.split(",")))
drv.run_local(range(2))
torch.cuda.synchronize()
print("set-up + 2 warm-up simulations: %.1f s" % (time.perf_counter() - t0), flush=True)
t0 = time.perf_counter()
drv.run_local(range(2, 2 + a.nsims))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.nsims
drv.run_local(range(100, 102), stage_times=True)
print("%dx%d %s: %.2f ms per simulation = %.1f sims/s; stages (ms): %s" % (a.side, a.side, a.prec, dt * 1e3, 1.0 / dt,
      ", ".join("%s %.2f" % kv for kv in drv.stage_ms.items())), flush=True)
if a.profile:
    pr = cProfile.Profile()
    pr.enable()
    drv.run_local(range(200, 203))
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
