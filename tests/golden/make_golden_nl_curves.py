#!/usr/bin/env python3
"""Reference-held lensing-noise curves as a small fixture (run in the build container only).

The reference's data/ directory holds N_L^kk curves whose generating configuration is not in the tree
(data/so_v3_1_deproj0_goal_fsky0p4_it.dat: columns named in its header; data/legacy/test_mv.csv: one MV curve).  They are
the only reference-held numbers that speak to the quadratic estimator's normalisation, so a sanity test compares
`lensing.NlGenerator` with their level, ordering and shape.  This script samples them at 64 multipoles into
nl_reference_curves.npz next to it (data only).

    python tests/golden/make_golden_nl_curves.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/data"


def main():
    so = np.loadtxt(os.path.join(REF, "so_v3_1_deproj0_goal_fsky0p4_it.dat"))
    mv = np.loadtxt(os.path.join(REF, "legacy", "test_mv.csv"))
    ells = np.unique(np.round(np.geomspace(10, 3000, 64))).astype(int)
    rows = so[np.searchsorted(so[:, 0], ells)]
    assert np.array_equal(rows[:, 0].astype(int), ells)
    mrows = mv[np.searchsorted(mv[:, 0], np.minimum(ells, int(mv[-1, 0])))]
    np.savez(os.path.join(HERE, "nl_reference_curves.npz"), ells=ells.astype(np.float64),
             so_columns=np.array(["TT", "TE", "EE", "TB", "EB", "Pol", "MV"]), so_nl=rows[:, 1:8],
             legacy_mv_ells=mrows[:, 0], legacy_mv=mrows[:, 1])
    print("wrote nl_reference_curves.npz:", ells.size, "multipoles")


if __name__ == "__main__":
    main()
