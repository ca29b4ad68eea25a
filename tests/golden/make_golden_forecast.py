#!/usr/bin/env python3
"""Golden vectors for the Gaussian bandpower covariance (run in the build container only).

`import orphics.cosmology` needs camb / pyfisher, but `LensForecast.loadKK / loadGenericCls / _bin_cls / KnoxCov /
sigmaClSquared / sn` (/root/reference/orphics/cosmology.py:976-1094) are NumPy + scipy.interpolate.interp1d around the
caller-supplied `theory` object.  Their definitions are taken out of the reference file with `ast` and executed as they
stand on a data-defined theory (linear interpolation of the C_ell tables stored in the fixture, either letter order).
Inputs + outputs go to forecast_reference.npz next to this script.  The fixture is data; no reference source travels.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_forecast.py
"""
import ast
import os
import types

import numpy as np
from scipy.interpolate import interp1d

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/orphics/cosmology.py"
NAMES = ["loadKK", "loadGenericCls", "_bin_cls", "KnoxCov", "sigmaClSquared", "sn"]


class TableTheory(object):
    def __init__(self):
        self.tab = {}

    def loadGenericCls(self, ells, cls, key, lpad=None):
        self.tab[key] = (np.asarray(ells, dtype=float), np.asarray(cls, dtype=float))

    def gCl(self, key, ells):
        e, c = self.tab[key] if key in self.tab else self.tab[key[::-1]]
        return np.interp(np.asarray(ells, dtype=float), e, c)


def main():
    tree = ast.parse(open(SRC).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "LensForecast"][0]
    ns = {"np": np, "interp1d": interp1d}
    for node in cls.body:
        if isinstance(node, ast.FunctionDef) and node.name in NAMES:
            exec(compile(ast.Module(body=[node], type_ignores=[]), SRC, "exec"), ns)
    me = types.SimpleNamespace(theory=TableTheory(), Nls={})
    for name in NAMES:
        setattr(me, name, types.MethodType(ns[name], me))
    ells = np.arange(2, 3001, dtype=float)
    out = {"ells": ells,
           "kk": 1e-7 * (ells / 100.) ** -1.2, "n_kk": 2e-8 * (1 + (ells / 1500.) ** 2),
           "gg": 3e-6 * (ells / 100.) ** -0.8, "n_gg": np.full(ells.shape, 4e-8),
           "kg": 2e-7 * (ells / 100.) ** -1.0,
           "edges": np.arange(100, 2000, 150), "fsky": np.float64(0.4)}
    me.loadKK(ells, out["kk"], ells, out["n_kk"])
    me.loadGenericCls("gg", ells, out["gg"], ells, out["n_gg"])
    me.loadGenericCls("kg", ells, out["kg"])
    for xy, wz in (("kk", "kk"), ("kg", "kg"), ("kk", "kg"), ("gg", "kk")):
        for ntot in (False, True):
            cov, s1, s2 = me.KnoxCov(xy, wz, out["edges"], 0.4, ntot=ntot)
            tag = "%s_%s_%d" % (xy, wz, int(ntot))
            out["cov_" + tag], out["s1_" + tag], out["s2_" + tag] = cov, s1, s2
    sn, errs = me.sn(out["edges"], 0.4, "kg")
    out["sn_kg"], out["errs_kg"] = np.float64(sn), errs
    out["sigma2_kk"] = me.sigmaClSquared("kk", out["edges"], 0.4)
    np.savez_compressed(os.path.join(HERE, "forecast_reference.npz"), **out)
    print("wrote forecast_reference.npz with %d arrays" % len(out))


if __name__ == "__main__":
    main()
