#!/usr/bin/env python3
"""Build the compact C_ell table shipped as orphics_amd/data/cosmo2017_cls.npz
from the reference's CAMB outputs (data files = inputs, not source).

Convention of cosmology.py:892-898 / default_theory (cosmology.py:850-852):
C_ell = D_ell * 2 pi / (ell (ell+1)), muK^2 (get_dimensionless=False),
kk from the lenspotential file: C^kk = 2 pi C^dd_col5 / 4 (cosmology.py:906-907).
Run in the build container only:  python tests/golden/make_theory_table.py
"""
import os
import numpy as np

REF = "/root/reference/data/cosmo2017_10K_acc3"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "orphics_amd", "data", "cosmo2017_cls.npz")

ell, ltt, lee, lbb, lte = np.loadtxt(REF + "_lensedCls.dat", unpack=True, usecols=[0, 1, 2, 3, 4])
lf = 2. * np.pi / ell / (ell + 1.)
uell, utt, uee, ute = np.loadtxt(REF + "_scalCls.dat", unpack=True, usecols=[0, 1, 2, 3])
uf = 2. * np.pi / uell / (uell + 1.)
elldd, cldd = np.loadtxt(REF + "_lenspotentialCls.dat", unpack=True, usecols=[0, 5])
clkk = 2. * np.pi * cldd / 4.
np.savez_compressed(OUT, l_ell=ell, l_TT=ltt * lf, l_EE=lee * lf, l_BB=lbb * lf, l_TE=lte * lf,
                    u_ell=uell, u_TT=utt * uf, u_EE=uee * uf, u_TE=ute * uf, kk_ell=elldd, kk=clkk)
print("wrote", os.path.abspath(OUT), os.path.getsize(OUT))
