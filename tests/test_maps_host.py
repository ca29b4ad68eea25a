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


def test_downsample_power_block_means_and_mapgen_ndown():
    """maps.py:1501-1550 / 1568-1569: block averaging + re-interpolation of a 2-D spectrum; closed forms: a constant
    spectrum is a fixed point (inside the coarse grid), order-0 sampling returns the block means piecewise, ``exp``
    acts on the block means, ``ndown < 1`` is the identity; MapGen(ndown=...) uses it for its covariance square root."""
    from orphics_amd import maps
    shape = (32, 64)
    g = _geom(shape)
    rng = np.random.default_rng(5)
    cov = rng.uniform(1.0, 2.0, (1, 1) + shape)
    assert maps.downsample_power(shape, g, cov, ndown=0) is cov
    flat = np.full((1, 1) + shape, 3.0)
    out = maps.downsample_power(shape, g, flat, ndown=4, order=0)
    sh_out = np.fft.fftshift(out, axes=(-2, -1))[0, 0]
    assert out.shape == flat.shape and np.allclose(sh_out[:29, :57], 3.0)       # (samples beyond the last coarse cell centre: see below)
    # one factor: the longer axis gets ndown * nmax / nmin (maps.py:1512-1518): here (4, 8); order 0 = nearest coarse cell
    got = np.fft.fftshift(maps.downsample_power(shape, g, cov, ndown=4, order=0), axes=(-2, -1))[0, 0]
    sh = np.fft.fftshift(cov, axes=(-2, -1))[0, 0]
    means = sh.reshape(8, 4, 8, 8).mean(axis=(1, 3))
    yy, xx = np.meshgrid(np.arange(32) / 4., np.arange(64) / 8., indexing="ij")
    iy, ix = np.floor(yy + 0.5).astype(int), np.floor(xx + 0.5).astype(int)      # nearest coarse sample, halves up
    inside = (yy <= 7) & (xx <= 7)             # scipy's 'constant' border: coordinates beyond the last sample are outside
    np.testing.assert_allclose(got[inside], means[iy[inside], ix[inside]], rtol=1e-13)
    assert np.all(got[~inside] == 0)                                   # sampled beyond the coarse grid: zero
    iy, ix = np.minimum(iy, 7), np.minimum(ix, 7)
    # exp acts on the block means
    sq = np.fft.fftshift(maps.downsample_power(shape, g, cov, ndown=(4, 8), order=0, exp=0.5), axes=(-2, -1))[0, 0]
    np.testing.assert_allclose(sq[inside], np.sqrt(means)[iy[inside], ix[inside]], rtol=1e-13)
    # linear order interpolates between neighbouring block means
    lin = np.fft.fftshift(maps.downsample_power(shape, g, cov, ndown=(4, 8), order=1), axes=(-2, -1))[0, 0]
    assert abs(lin[2, 4] - (means[0, 0] * 0.25 + means[1, 0] * 0.25 + means[0, 1] * 0.25 + means[1, 1] * 0.25)) < 1e-12
    mg = maps.MapGen(shape, g, cov * 1e-3, ndown=4, order=0)
    want = maps.downsample_power(shape, g, cov * 1e-3 * (32 * 64) / g.area, 4, 0, exp=0.5)
    np.testing.assert_allclose(mg.covsqrt, want, rtol=1e-13)
