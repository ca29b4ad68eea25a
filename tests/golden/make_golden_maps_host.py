#!/usr/bin/env python3
"""Golden vectors for the PURE-NumPy host helpers of the reference's maps.py (run in the build container only).

`import orphics.maps` fails here (pixell is absent), but `gauss_beam`, `cosine_window`, `kspace_coadd`, the body of
`FourierCalc.f2power`, the multi-component branch of `FourierCalc.power2d` and `lensing.fkappa_to_fphi` do not touch pixell: their function definitions are taken out of /root/reference/orphics/maps.py
with `ast` and executed as they stand (no stand-in for any missing module is written), on seeded inputs; inputs + outputs are stored in
maps_host_reference.npz next to this script.  The fixture is data; no reference source travels.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_maps_host.py
"""
import ast
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/orphics/maps.py"


def reference_functions(names):
    tree = ast.parse(open(SRC).read())
    ns = {"np": np}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), SRC, "exec"), ns)
    missing = [n for n in names if n not in ns]
    assert not missing, missing
    return ns


def reference_method(cls, name):
    tree = ast.parse(open(SRC).read())
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name == name:
                    ns = {"np": np}
                    exec(compile(ast.Module(body=[sub], type_ignores=[]), SRC, "exec"), ns)
                    return ns[name]
    raise KeyError((cls, name))


def main():
    import types
    ns = reference_functions(["gauss_beam", "cosine_window", "kspace_coadd"])
    out = {}
    rng = np.random.default_rng(3)
    ell = np.abs(rng.standard_normal((16, 24))) * 3000.
    for i, fwhm in enumerate((1.5, 7.0)):
        out["beam_ell"] = ell
        out["beam_fwhm_%d" % i] = np.float64(fwhm)
        out["beam_out_%d" % i] = ns["gauss_beam"](ell, fwhm)
    cases = [(64, 80, 10, 12, 3, 2), (32, 32, 30, 30, 0, 0), (20, 16, 12, 9, 2, 3), (50, 40, 0, 7, 0, 1), (40, 40, 5, 0, 2, 0), (16, 16, 30, 30, 1, 1)]
    out["win_cases"] = np.array(cases)
    for i, (Ny, Nx, ay, ax, py, px) in enumerate(cases):
        out["win_out_%d" % i] = ns["cosine_window"](Ny, Nx, lenApodY=ay, lenApodX=ax, padY=py, padX=px)
    # FourierCalc.f2power's body (maps.py:1620-1624) is one NumPy expression of its arguments
    k1 = rng.standard_normal((12, 10)) + 1j * rng.standard_normal((12, 10))
    k2 = rng.standard_normal((12, 10)) + 1j * rng.standard_normal((12, 10))
    out["f2_k1"], out["f2_k2"], out["f2_norm"] = k1, k2, np.float64(0.37)
    f2power = reference_method("FourierCalc", "f2power")
    fake_self = types.SimpleNamespace(normfact=0.37)
    out["f2_out"] = f2power(fake_self, k1, k2)
    out["f2_out_pixel_units"] = f2power(fake_self, k1, k2, pixel_units=True)
    # FourierCalc.power2d (maps.py:1639-1677) with both transforms supplied (kmap, kmap2): the (ncomp, ncomp, Ny, Nx) assembly
    # -- autos on the diagonal, upper-triangle crosses mirrored, skip_cross, pixel_units -- touches no pixell function
    class WithWcs(np.ndarray):
        wcs = None
    power2d = reference_method("FourierCalc", "power2d")
    fake_self.f2power = types.MethodType(f2power, fake_self)
    ka = (rng.standard_normal((3, 32, 36)) + 1j * rng.standard_normal((3, 32, 36)))
    kb = (rng.standard_normal((3, 32, 36)) + 1j * rng.standard_normal((3, 32, 36)))
    out["p2d_k1"], out["p2d_k2"] = ka, kb
    out["p2d_cross"] = power2d(fake_self, kmap=ka.copy().view(WithWcs), kmap2=kb.copy().view(WithWcs))[0]
    out["p2d_auto"] = power2d(fake_self, kmap=ka.copy().view(WithWcs))[0]
    out["p2d_skip_cross_pixel_units"] = power2d(fake_self, kmap=ka.copy().view(WithWcs), kmap2=kb.copy().view(WithWcs), skip_cross=True,
                                                pixel_units=True)[0]
    # kspace_coadd (maps.py:1098-1114), with zero-noise and zero-beam modes to exercise the non-finite handling
    km = rng.standard_normal((3, 32, 36)) + 1j * rng.standard_normal((3, 32, 36))
    kb = rng.uniform(0.2, 1.0, (3, 32, 36))
    kn = rng.uniform(0.5, 2.0, (3, 32, 36))
    kn[0, 2, 3] = 0.0
    kn[:, 5, 5] = 0.0
    kb[:, 7, 1] = 0.0
    with np.errstate(divide="ignore", invalid="ignore"):
        out["coadd_kmaps"], out["coadd_kbeams"], out["coadd_kncovs"] = km, kb, kn
        out["coadd_out"] = ns["kspace_coadd"](km.copy(), kb.copy(), kn.copy(), fkbeam=0.8)
    # lensing.fkappa_to_fphi (lensing.py:662-665): phi = 2 kappa / (l (l + 1)), zero below l = 2
    global SRC
    SRC_MAPS = SRC
    SRC = "/root/reference/orphics/lensing.py"
    f2p = reference_functions(["fkappa_to_fphi"])["fkappa_to_fphi"]
    SRC = SRC_MAPS
    ml = np.abs(rng.standard_normal((12, 10))) * 500.
    ml[0, 0] = 0.0
    ml[1, 1] = 1.5
    fk = rng.standard_normal((12, 10)) + 1j * rng.standard_normal((12, 10))
    with np.errstate(divide="ignore", invalid="ignore"):
        out["fphi_modlmap"], out["fphi_fkappa"], out["fphi_out"] = ml, fk, f2p(fk.copy(), ml.copy())
    np.savez_compressed(os.path.join(HERE, "maps_host_reference.npz"), **out)
    print("wrote maps_host_reference.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
