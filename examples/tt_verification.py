#!/usr/bin/env python3
"""The reference's tutorials/tt_verification.ipynb (cells 1-5) with `orphics` replaced by `orphics_amd`:
lensed CMB simulations on a 10 deg patch at 0.5' (1200 x 1200 pixels, pol=True), TT and EB quadratic estimators,
cross-power of the reconstruction with the input kappa, mean fractional difference from the input auto-power.

    python examples/tt_verification.py [Nsims] [width_deg]

Only the import line and `wcs` (a FlatGeometry here) differ from the notebook; plotting is replaced by a table.
width_deg = 8.5333 gives a 1024^2 patch (power-of-two sides -> fused kernels)."""
from __future__ import print_function

import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from orphics_amd import cosmology, lensing, maps, stats

Nsims = int(sys.argv[1]) if len(sys.argv) > 1 else 20
width_deg = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0

# --- cell 1
shape, wcs = maps.rect_geometry(width_deg=width_deg, px_res_arcmin=0.5)
shape = (3,) + shape
theory = cosmology.default_theory()
beam_arcmin = 1.5
noise_uk_arcmin = 1.0
noisep = noise_uk_arcmin * np.sqrt(2.)
flsims = lensing.FlatLensingSims(shape, wcs, theory, beam_arcmin, noise_uk_arcmin, noise_e_uk_arcmin=noisep,
                                 noise_b_uk_arcmin=noisep, pol=True, fixed_lens_kappa=None)

# --- cell 3
n2d = np.nan_to_num(flsims.ps_noise[0, 0])
n2p = np.nan_to_num(flsims.ps_noise[1, 1])
tellmin, tellmax = 300, 2000
pellmin, pellmax = 300, 2000
kellmin, kellmax = 20, 3500
tmask = maps.mask_kspace(shape, wcs, lmin=tellmin, lmax=tellmax)
pmask = maps.mask_kspace(shape, wcs, lmin=pellmin, lmax=pellmax)
kmask = maps.mask_kspace(shape, wcs, lmin=kellmin, lmax=kellmax)
t0 = time.time()
qest = lensing.qest(shape, wcs, theory, noise2d=n2d, beam2d=flsims.kbeam, kmask=tmask, noise2d_P=n2p, kmask_P=pmask,
                    kmask_K=kmask, pol=True, grad_cut=None, unlensed_equals_lensed=True, bigell=9000)
print("estimator set up for %s in %.1f s" % (str(shape), time.time() - t0))

# --- cell 4
fc = maps.FourierCalc(shape, wcs)
nbins = 20
bin_edges = np.linspace(kellmin, kellmax, nbins)
binner = stats.bin2D(flsims.modlmap, bin_edges)
st = stats.Stats()
t0 = time.time()
for i in range(Nsims):
    unlensed, kappa, lensed, beamed, noise_map, observed = flsims.get_sim(return_intermediate=True)
    _, kmapTEB, _ = fc.power2d(observed)
    recon = qest.kappa_from_map("TT", kmapTEB[0], alreadyFTed=True)
    pcross, _, _ = fc.power2d(recon, kappa)
    reconEB = qest.kappa_from_map("EB", kmapTEB[0], kmapTEB[1], kmapTEB[2], alreadyFTed=True)
    pcrossEB, _, _ = fc.power2d(reconEB, kappa)
    pii, _, _ = fc.power2d(kappa)
    cents, p1d = binner.bin(pcross)
    cents, p1dEB = binner.bin(pcrossEB)
    cents, pii1d = binner.bin(pii)
    st.add_to_stats("ratio", (p1d - pii1d) / pii1d)
    st.add_to_stats("ratioEB", (p1dEB - pii1d) / pii1d)
print("%d simulations + TT and EB reconstructions in %.1f s" % (Nsims, time.time() - t0))
st.get_stats()

# --- cell 5 (table instead of the plot)
y, yerr = st.stats['ratio']['mean'], st.stats['ratio']['errmean']
yEB, yerrEB = st.stats['ratioEB']['mean'], st.stats['ratioEB']['errmean']
print("%8s %12s %10s %12s %10s" % ("L", "dC/C (TT)", "+-", "dC/C (EB)", "+-"))
for a, b, c, d, e in zip(cents, y, yerr, yEB, yerrEB):
    print("%8.0f %12.4f %10.4f %12.4f %10.4f" % (a, b, c, d, e))
