#!/usr/bin/env python3
"""cProfile of the five-estimator set-up (lensing.qest(pol=True) + MV weights) at N^2: where the host time of BASELINE config 3's
set-up goes.  usage: python tools/profile_setup.py [N]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from orphics_amd import cosmology, lensing, maps  # noqa: E402
from orphics_amd.geometry import FlatGeometry  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
shape = (N, N)
g = FlatGeometry.from_res(shape, 0.5)
th = cosmology.default_theory()


def run():
    ml = g.modlmap()
    beam = maps.gauss_beam(ml, 1.5)
    noise = np.full(shape, cosmology.white_noise_power(1.0))
    tmask = maps.mask_kspace(shape, g, lmin=300, lmax=2000)
    kmask = maps.mask_kspace(shape, g, lmin=20, lmax=3500)
    q = lensing.qest(shape, g, th, noise2d=noise, beam2d=beam, kmask=tmask, noise2d_P=2 * noise, kmask_P=tmask, kmask_K=kmask, pol=True,
                     unlensed_equals_lensed=True, dtype="f64")
    e = q.eng
    k = [e.grf_hc(1, i) for i in range(3)]
    out = q.reconstruct_mv_hc(*k, estimators=("TT", "TE", "EE", "EB", "TB"))
    torch.cuda.synchronize()
    return out


t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
run()
pr.disable()
print("total %.1f s" % (time.perf_counter() - t0))
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(25)
