#!/usr/bin/env python3
"""Golden vectors for the split-based estimators (run in the build container only).

`SplitLensing.qpower / qfrag / cross_estimator` (/root/reference/orphics/lensing.py:966-1003) and `split_calc`
(/root/reference/orphics/maps.py:2296-2333) are pure NumPy around two caller-supplied objects: a `qest` with a
`kappa_from_map` method and a `FourierCalc` with `f2power`.  Their definitions are taken out of the reference files with
`ast` and executed as they stand.  The caller-supplied objects are data-defined here:

  * `qest.kappa_from_map(XY, T2DData=a, T2DDataY=b, ...)` = U * a * b + V * a * roll(b, (1, 2))  -- an arbitrary bilinear
    map of the two legs given by the seeded complex planes U, V stored in the fixture (the tests rebuild it from them);
  * `fc.f2power` is the reference's own `FourierCalc.f2power` body with `normfact` from the fixture.

Inputs + outputs go to splits_reference.npz next to this script.  The fixture is data; no reference source travels.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_splits.py
"""
import ast
import os
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/orphics"


def _defs(path, names, cls=None):
    tree = ast.parse(open(path).read())
    body = tree.body
    if cls is not None:
        body = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls][0].body
    ns = {"np": np}
    for node in body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    missing = [n for n in names if n not in ns]
    assert not missing, missing
    return ns


class WithWcs(np.ndarray):
    """split_calc reads `.wcs` off its first argument (and never uses it when a fourier_calc is passed)."""
    wcs = None


def bilinear_qest(U, V):
    def kappa_from_map(XY, T2DData=None, T2DDataY=None, alreadyFTed=True, returnFt=True, **unused):
        assert XY == "TT" and alreadyFTed and returnFt
        return U * T2DData * T2DDataY + V * T2DData * np.roll(T2DDataY, (1, 2), (0, 1))
    return types.SimpleNamespace(kappa_from_map=kappa_from_map)


def main():
    rng = np.random.default_rng(11)
    shape = (32, 36)            # the product's engines need even sides >= 32
    cplx = lambda *lead: rng.standard_normal(lead + shape) + 1j * rng.standard_normal(lead + shape)   # noqa: E731
    out = {}
    normfact = 0.83
    f2power = _defs(REF + "/maps.py", ["f2power"], cls="FourierCalc")["f2power"]
    fc = types.SimpleNamespace(normfact=normfact)
    fc.f2power = types.MethodType(f2power, fc)
    out["normfact"] = np.float64(normfact)

    # SplitLensing.cross_estimator for 4, 5 and 6 splits
    sl_ns = _defs(REF + "/lensing.py", ["qpower", "qfrag", "cross_estimator"], cls="SplitLensing")
    U, V = cplx(), cplx()
    out["U"], out["V"] = U, V
    for n in (4, 5, 6):
        splits = cplx(n) + 3.0 * cplx()[None]          # common signal + independent noise
        me = types.SimpleNamespace(fc=fc, qest=bilinear_qest(U, V), est="TT")
        for name in ("qpower", "qfrag", "cross_estimator"):
            setattr(me, name, types.MethodType(sl_ns[name], me))
        out["cross_splits_%d" % n] = splits
        out["cross_out_%d" % n] = me.cross_estimator(splits.copy())

    # split_calc, both branches
    sc = _defs(REF + "/maps.py", ["split_calc"])["split_calc"]
    isp, jsp = cplx(4), cplx(4)
    ico, jco = isp.mean(0), jsp.mean(0)
    out["sc_isplits"], out["sc_jsplits"] = isp, jsp
    for alt in (True, False):
        t, c, nz = sc(isp.copy().view(WithWcs), jsp.copy().view(WithWcs), ico.copy(), jco.copy(), fourier_calc=fc, alt=alt)
        tag = "alt" if alt else "loop"
        out["sc_total_" + tag], out["sc_crosses_" + tag], out["sc_noise_" + tag] = np.asarray(t), np.asarray(c), np.asarray(nz)
    np.savez_compressed(os.path.join(HERE, "splits_reference.npz"), **out)
    print("wrote splits_reference.npz with %d arrays" % len(out))


if __name__ == "__main__":
    main()
