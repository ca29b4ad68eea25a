"""Host-side pieces of orphics_amd.maps that need no GPU: spec2flat / smooth_spectrum (MapGen's 3-D covariance
path, maps.py:1573) against the oracle's independent restatement and closed forms."""
import numpy as np

from oracle import maps_oracle as mo


def _geom(shape, res=2.0):
    from orphics_amd.geometry import FlatGeometry
    return FlatGeometry.from_res(shape, res)


def test_spec2flat_scalar_matches_oracle_and_closed_form():
    from orphics_amd import maps
    shape = (48, 64)
    g = _geom(shape)
    ell = np.arange(9000.)
    cl = 2e3 / (1 + (ell / 300.) ** 2.5)
    got = maps.spec2flat(shape, g, cl[None, None], 0.5, smooth=0)
    want = mo.spec2flat(shape, g.step_y, g.step_x, cl[None, None], 0.5)
    assert got.shape == (1, 1) + shape
    np.testing.assert_allclose(got, want, rtol=1e-13)
    # closed form: the SQUARE ROOT sqrt(C_l Npix / area) is what gets interpolated (power first, then interpolation)
    ml = g.modlmap()
    np.testing.assert_allclose(got[0, 0], np.interp(ml, ell, np.sqrt(cl * (48 * 64 / g.area))), rtol=1e-13)
    np.testing.assert_allclose(got[0, 0] ** 2, np.interp(ml, ell, cl) * (48 * 64 / g.area), rtol=1e-5)
    # beyond the table: zero ("constant" border)
    short = maps.spec2flat(shape, g, cl[None, None, :2000], 0.5, smooth=0)
    assert np.all(short[0, 0][ml > 1999] == 0) and np.any(short[0, 0][ml < 1999] > 0)


def test_spec2flat_matrix_sqrt_and_smoothing():
    from orphics_amd import maps
    shape = (32, 32)
    g = _geom(shape, 4.0)
    ell = np.arange(4000.)
    tt = 1e3 / (1 + (ell / 200.) ** 2)
    cov = np.zeros((3, 3, ell.size))
    cov[0, 0], cov[1, 1], cov[2, 2] = tt, 0.1 * tt, 0.01 * tt
    cov[0, 1] = cov[1, 0] = 0.2 * tt
    cs = maps.spec2flat((3,) + shape, g, cov, 0.5, smooth=0)
    np.testing.assert_allclose(cs, mo.spec2flat(shape, g.step_y, g.step_x, cov, 0.5), rtol=1e-10, atol=1e-12)
    ml = g.modlmap()
    sel = (ml > 10) & (ml < 3900)
    back = np.einsum("abyx,bcyx->acyx", cs, cs) / (32 * 32 / g.area)
    for i, j in ((0, 0), (0, 1), (1, 1), (2, 2)):
        # interpolating the matrix square root and squaring again is not exactly interpolating the spectrum
        np.testing.assert_allclose(back[i, j][sel], np.interp(ml, ell, cov[i, j])[sel], rtol=2e-3)
    # smoothing: "auto" width = mean fundamental / 3.41; a flat spectrum is a fixed point, the oracle agrees
    ly, lx = g.laxes()
    width = 0.5 * (abs(ly[1] - ly[0]) + abs(lx[1] - lx[0])) / 3.41
    flat = np.full((1, 1, 3000), 7.0)
    np.testing.assert_allclose(maps.smooth_spectrum(flat, width), flat, rtol=1e-12)
    sm = maps.spec2flat(shape, g, tt[None, None], 0.5, smooth="auto")
    np.testing.assert_allclose(sm, mo.spec2flat(shape, g.step_y, g.step_x, tt[None, None], 0.5, smooth_width=width), rtol=1e-10)
    assert 0 < np.abs(sm / maps.spec2flat(shape, g, tt[None, None], 0.5, smooth=0) - 1)[0, 0][sel].max() < 0.2
