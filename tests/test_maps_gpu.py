"""GPU parity of orphics_amd.maps / stats (through the C-ABI) vs the NumPy oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import maps_oracle as mo  # noqa: E402
from oracle import stats_oracle as so  # noqa: E402

RES = 2.0


def geom(shape, res=RES):
    from orphics_amd.geometry import FlatGeometry
    return FlatGeometry.from_res(shape, res)


def rel(a, b):
    return np.abs(np.asarray(a) - b).max() / np.abs(b).max()


@pytest.mark.parametrize("dt,tol", [(np.float64, 1e-12), (np.float32, 5e-6)])
def test_fouriercalc_scalar(dt, tol):
    from orphics_amd import maps
    shape = (128, 256)
    g = geom(shape)
    fc = maps.FourierCalc(shape, g)
    fo = mo.FourierCalc(shape, g.step_y, g.step_x)
    assert fc.normfact == fo.normfact
    rng = np.random.default_rng(0)
    m1 = rng.standard_normal(shape).astype(dt)
    m2 = rng.standard_normal(shape).astype(dt)
    assert rel(fc.fft(m1), fo.fft(m1)) < tol
    p, k1, k2 = fc.power2d(m1, m2)
    po, k1o, k2o = fo.power2d(m1.astype(np.float64), m2.astype(np.float64))
    assert isinstance(p, np.ndarray) and p.shape == shape and k1.shape == shape
    assert rel(p, po) < 20 * tol and rel(k1, k1o) < tol and rel(k2, k2o) < tol
    p2, _, _ = fc.power2d(kmap=k1, kmap2=k2)
    assert rel(p2, po) < 20 * tol
    pp, kk = fc.f1power(m1, k2)
    assert rel(pp, po) < 20 * tol
    assert rel(fc.f2power(k1, k2, pixel_units=True), fo.f2power(k1o, k2o, pixel_units=True)) < 20 * tol
    z = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex128 if dt == np.float64 else np.complex64)
    assert rel(fc.ifft(z), fo.ifft(z.astype(np.complex128))) < tol
    assert rel(fc.iqu2teb(m1, normalize=True), fo.iqu2teb(m1.astype(np.float64), normalize=True)) < tol


def test_fouriercalc_pol_and_half_layout():
    from orphics_amd import maps
    from orphics_amd.stats import HalfPlane
    shape = (3, 64, 128)
    g = geom(shape)
    rng = np.random.default_rng(1)
    m = rng.standard_normal(shape)
    for iau in (False, True):
        fc = maps.FourierCalc(shape, g, iau=iau)
        fo = mo.FourierCalc(shape, g.step_y, g.step_x, iau=iau)
        p, k1, _ = fc.power2d(m)
        po, k1o, _ = fo.power2d(m)
        assert p.shape == (3, 3, 64, 128)
        assert rel(k1, k1o) < 1e-12 and rel(p, po) < 1e-11
        ps, _, _ = fc.power2d(m, skip_cross=True)
        assert np.all(ps[0, 1] == 0) and rel(ps[1, 1], po[1, 1]) < 1e-11
    fh = maps.FourierCalc(shape, g, layout="half")
    ph, kh, _ = fh.power2d(m)
    assert isinstance(ph, HalfPlane) and isinstance(kh, HalfPlane)
    # half layout follows the Hermitian convention at the self-conjugate Nyquist row/column
    # (where the reference's full-plane E/B are not Hermitian); everything else is identical
    pref = mo.FourierCalc(shape, g.step_y, g.step_x).power2d(m)[0]
    got = ph.numpy()
    keep = np.ones(shape[-2:], bool); keep[shape[-2] // 2, :] = False; keep[:, shape[-1] // 2] = False
    assert rel(got[..., keep], pref[..., keep]) < 1e-11
    back = fh.ifft(kh[0]).cpu().numpy()
    assert rel(back, m[0]) < 1e-12


def test_filter_map_beam_mask():
    from orphics_amd import maps
    shape = (128, 128)
    g = geom(shape)
    ml = g.modlmap()
    rng = np.random.default_rng(2)
    m = rng.standard_normal((2,) + shape)
    kb = maps.gauss_beam(ml, 1.5)
    assert np.array_equal(kb, mo.gauss_beam(ml, 1.5))
    assert rel(maps.filter_map(m, kb), mo.filter_map(m, kb)) < 1e-12
    assert np.allclose(maps.filter_map(m, np.ones(shape)), m, atol=1e-13)
    for kw in (dict(lmin=300, lmax=2000), dict(lxcut=90, lycut=50), dict(lmin=100, lxcut=20)):
        a = maps.mask_kspace(shape, g, **kw)
        b = mo.mask_kspace(shape, g.step_y, g.step_x, **kw)
        assert a.dtype == b.dtype and np.array_equal(a, b)
    km = maps.mask_kspace(shape, g, lmin=300, lmax=2000)
    assert rel(maps.filter_map(m[0], km), mo.filter_map(m[0], km)) < 1e-12
    # a non-symmetric real filter takes the general (C2C) path
    f = rng.uniform(0, 1, shape)
    assert rel(maps.filter_map(m[0], f), mo.filter_map(m[0], f)) < 1e-12
    t, w2 = maps.get_taper(shape, g)
    to, w2o = mo.get_taper(shape)
    assert np.array_equal(t, to) and w2 == w2o


def test_filter_map_complex_filter_is_a_shift():
    """maps.filter_map with a COMPLEX k-space filter (maps.py:1923 takes any array): the phase ramp exp(-i l.dx) of a
    whole-pixel shift reproduces np.roll; a general complex filter equals the NumPy expression Re(IFFT(FFT(m) F))."""
    from orphics_amd import maps
    from orphics_amd.geometry import FlatGeometry
    ny, nx = 128, 256
    rng = np.random.default_rng(9)
    m = rng.standard_normal((ny, nx))
    ky = np.fft.fftfreq(ny)[:, None]; kx = np.fft.fftfreq(nx)[None, :]
    shift = np.exp(-2j * np.pi * (3 * ky + 5 * kx))
    out = maps.filter_map(m, shift)
    assert np.abs(out - np.roll(m, (3, 5), axis=(0, 1))).max() < 1e-10
    F = rng.standard_normal((ny, nx)) + 1j * rng.standard_normal((ny, nx))
    want = np.real(np.fft.ifft2(np.fft.fft2(m) * F))
    assert np.abs(maps.filter_map(m, F) - want).max() < 1e-10 * np.abs(want).max()
    out32 = maps.filter_map(m.astype(np.float32), shift)
    assert out32.dtype == np.float32 and np.abs(out32 - np.roll(m, (3, 5), axis=(0, 1))).max() < 2e-5


def test_mapgen_parity_and_statistics():
    from orphics_amd import maps
    shape = (128, 128)
    g = geom(shape)
    ml = g.modlmap()
    cov = (1.0 / (1 + (ml / 500.) ** 2)).reshape((1, 1) + shape)
    mg = maps.MapGen(shape, g, cov, dtype="f64")
    mgo = mo.MapGen(shape, g.step_y, g.step_x, cov)
    assert np.allclose(mg.covsqrt, mgo.covsqrt)
    rng = np.random.default_rng(3)
    rand = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    assert rel(mg.get_map_from_rand(rand, scalar=True), mgo.get_map_from_rand(rand, scalar=True)) < 1e-12
    # device draws: power of the realisations matches the input spectrum
    fo = mo.FourierCalc(shape, g.step_y, g.step_x)
    acc = 0
    for s in range(24):
        acc = acc + fo.power2d(mg.get_map(seed=s, scalar=True).cpu().numpy())[0]
    ratio = (acc / 24)[ml > 0] / cov[0, 0][ml > 0]
    assert abs(ratio.mean() - 1) < 0.02
    a = mg.get_map(seed=5, scalar=True); b = mg.get_map(seed=5, scalar=True); c = mg.get_map(seed=6, scalar=True)
    assert torch.equal(a, b) and not torch.equal(a, c)
    # polarised: TT/EE/TE spectra of the (T,Q,U) draw
    shape3 = (3,) + shape
    tt = 1.0 / (1 + (ml / 500.) ** 2); ee = 0.1 * tt; te = 0.15 * tt
    cov3 = np.zeros((3, 3) + shape); cov3[0, 0] = tt; cov3[1, 1] = ee; cov3[0, 1] = cov3[1, 0] = te; cov3[2, 2] = 0.01 * tt
    mg3 = maps.MapGen(shape3, geom(shape3), cov3, dtype="f64")
    fo3 = mo.FourierCalc(shape3, g.step_y, g.step_x)
    acc = 0
    for s in range(24):
        acc = acc + fo3.power2d(mg3.get_map(seed=100 + s).cpu().numpy())[0]
    acc /= 24
    sel = (ml > 200) & (ml < 3000)
    assert abs(acc[0, 0][sel].sum() / tt[sel].sum() - 1) < 0.03
    assert abs(acc[1, 1][sel].sum() / ee[sel].sum() - 1) < 0.03
    assert abs(acc[0, 1][sel].sum() / te[sel].sum() - 1) < 0.06
    assert abs(acc[2, 2][sel].sum() / (0.01 * tt)[sel].sum() - 1) < 0.03
    rand3 = rng.standard_normal(shape3) + 1j * rng.standard_normal(shape3)
    mgo3 = mo.MapGen(shape3, g.step_y, g.step_x, cov3)
    assert rel(mg3.get_map_from_rand(rand3), mgo3.get_map_from_rand(rand3)) < 1e-11


def test_mapgen_1d_spectra_and_real_space_draw():
    """MapGen from (ncomp,ncomp,lmax) spectra (spec2flat, maps.py:1573) and get_map(real=True) (maps.py:1578):
    covsqrt equals the oracle's restatement; binned power of the draws follows the input C_l for both draw modes
    (demo-grf.ipynb cell 7 criterion); real / harmonic draws are different realisations of the same field."""
    from orphics_amd import maps
    shape = (256, 256)
    g = geom(shape)
    ml = g.modlmap()
    ell = np.arange(7000.)
    cl = 5e3 / (1 + (ell / 400.) ** 2.2)
    mg = maps.MapGen(shape, g, cl[None, None], smooth=0, dtype="f64")
    np.testing.assert_allclose(mg.covsqrt, mo.spec2flat(shape, g.step_y, g.step_x, cl[None, None], 0.5), rtol=1e-12)
    fo = mo.FourierCalc(shape, g.step_y, g.step_x)
    edges = np.arange(200, 4000, 200.)
    b = so.bin2D(ml, edges)
    cents = (edges[1:] + edges[:-1]) / 2
    for real in (False, True):
        acc = 0
        for sd in range(16):
            acc = acc + fo.power2d(mg.get_map(seed=sd, scalar=True, real=real).cpu().numpy())[0]
        _, p1d = b.bin(acc / 16)
        assert np.abs(p1d / np.interp(cents, ell, cl) - 1).max() < 0.05
    a = mg.get_map(seed=3, scalar=True, real=True)
    assert torch.equal(a, mg.get_map(seed=3, scalar=True, real=True))
    assert not torch.equal(a, mg.get_map(seed=3, scalar=True, real=False))
    # default smoothing ("auto") changes the plane only at the few-per-cent level for a smooth spectrum
    mgs = maps.MapGen(shape, g, cl[None, None], dtype="f64")
    sel = (ml > 100) & (ml < 6000)
    assert 0 < np.abs(mgs.covsqrt[0, 0][sel] / mg.covsqrt[0, 0][sel] - 1).max() < 0.1
    # polarised 3-D covariance
    cov3 = np.zeros((3, 3, ell.size))
    cov3[0, 0], cov3[1, 1], cov3[2, 2] = cl, 0.1 * cl, 0.01 * cl
    cov3[0, 1] = cov3[1, 0] = 0.2 * cl
    shape3 = (3,) + shape
    mg3 = maps.MapGen(shape3, geom(shape3), cov3, smooth=0, dtype="f32")
    fo3 = mo.FourierCalc(shape3, g.step_y, g.step_x)
    acc = 0
    for sd in range(12):
        acc = acc + fo3.power2d(mg3.get_map(seed=50 + sd).cpu().numpy().astype(np.float64))[0]
    acc /= 12
    for (i, j), amp in (((0, 0), 1.0), ((1, 1), 0.1), ((0, 1), 0.2), ((2, 2), 0.01)):
        _, p1d = b.bin(acc[i, j])
        assert np.abs(p1d / (amp * np.interp(cents, ell, cl)) - 1).max() < (0.12 if i != j else 0.06)
    # ndown (maps.py:1568-1569): the 4-D covariance is block-averaged / re-interpolated before the square root; the draws
    # follow the smoothed spectrum (a smooth C_l is nearly a fixed point of the smoothing)
    cov4 = np.interp(g.modlmap(), ell, cl)[None, None]
    mgd = maps.MapGen(shape, g, cov4, ndown=2, order=1)
    np.testing.assert_allclose(mgd.covsqrt, maps.downsample_power(shape, g, cov4 * np.prod(shape) / g.area, 2, 1, exp=0.5), rtol=1e-13)
    accd = 0
    for sd in range(12):
        accd = accd + fo.power2d(mgd.get_map(seed=80 + sd).cpu().numpy().astype(np.float64))[0]
    _, pd1 = b.bin(accd / 12)
    mid = (cents > 0.15 * cents.max()) & (cents < 0.7 * cents.max())
    assert np.abs(pd1 / np.interp(cents, ell, cl) - 1)[mid].max() < 0.15


def test_bin2d_against_reference_golden(golden_dir):
    """Product bin2D (HIP) vs the fixtures produced by the real orphics.stats.bin2D."""
    from orphics_amd import stats
    g = np.load(os.path.join(golden_dir, "bin2d_reference.npz"))
    b = stats.bin2D(g["a_modlmap"], g["a_edges"])
    assert np.array_equal(b.digitized, g["a_digitized"])
    assert b.digitized.dtype == np.int64
    c, r, cnt = b.bin(g["a_data"], get_count=True)
    assert np.array_equal(c, g["a_cents"]) and np.array_equal(cnt, g["a_count"])
    np.testing.assert_allclose(r, g["a_res"], rtol=1e-13)
    np.testing.assert_allclose(b.bin(g["a_data"], weights=g["a_weights"])[1], g["a_res_w"], rtol=1e-13)
    _, r, cnt = b.bin(g["a_data_nan"], mask_nan=True, get_count=True)
    assert np.array_equal(cnt, g["a_count_nan"])
    np.testing.assert_allclose(r, g["a_res_nan"], rtol=1e-13)
    _, _, s = b.bin(g["a_data"], err=True)
    np.testing.assert_allclose(s, so.bin2D(g["a_modlmap"], g["a_edges"]).bin(g["a_data"], err=True)[2], rtol=1e-10)
    # the documented deviation: the reference's err loop is shifted by one bin (stats.py:799-801); the flag reproduces
    # the real reference's output (fixture), the default does not
    _, _, s_ref = b.bin(g["a_data"], err=True, err_reference_indexing=True)
    np.testing.assert_allclose(s_ref, g["a_err_ref_shifted"], rtol=1e-10)
    assert np.max(np.abs(s / g["a_err_ref_shifted"] - 1)) > 1e-3
    bt = stats.bin2D(g["t_modlmap"], g["t_edges"])
    assert np.array_equal(bt.digitized, g["t_digitized"])  # exact ties on integer edges
    _, r, cnt = bt.bin(g["t_data"], get_count=True)
    assert np.array_equal(cnt, g["t_count"])
    np.testing.assert_allclose(r, g["t_res"], rtol=1e-13)
    b3 = stats.bin2D(g["h3_modrmap"], g["h3_edges"])  # H3 quirk reproduced
    c3, r3 = b3.bin(g["h3_data"])
    np.testing.assert_allclose(r3, g["h3_res"])
    assert np.array_equal(stats.bin2D(g["tie_vals"], g["tie_edges"]).digitized, g["tie_digitized"])
    c, r = stats.bin_in_annuli(g["a_data"], g["a_modlmap"], g["a_edges"])
    np.testing.assert_allclose(r, g["a_res"], rtol=1e-13)


@pytest.mark.parametrize("N,res", [(1024, 2.0)])
def test_config1_power2d_bin2d(N, res):
    """BASELINE config 1: 1024^2 2' GRF auto-spectrum + bin2D, full- and half-plane paths."""
    from orphics_amd import maps, stats
    shape = (N, N)
    g = geom(shape, res)
    ml = g.modlmap()
    rng = np.random.default_rng(0)
    m = rng.standard_normal(shape)
    edges = np.arange(100, 3000, 40.)
    fo = mo.FourierCalc(shape, g.step_y, g.step_x)
    bo = so.bin2D(ml, edges)
    co, ro = bo.bin(fo.power2d(m)[0])
    binner = stats.bin2D(ml, edges)
    assert np.array_equal(binner.digitized, bo.digitized)
    for dt, tol in ((np.float64, 1e-12), (np.float32, 1e-5)):
        fc = maps.FourierCalc(shape, g)
        p2d, _, _ = fc.power2d(m.astype(dt))
        c, r, cnt = binner.bin(p2d, get_count=True)
        assert np.array_equal(cnt, bo.bin(m, get_count=True)[2])
        assert np.max(np.abs(r / ro - 1)) < tol
        fh = maps.FourierCalc(shape, g, layout="half")
        ph, _, _ = fh.power2d(torch.as_tensor(m.astype(dt)).cuda())
        c2, r2, cnt2 = binner.bin(ph, get_count=True)
        assert np.array_equal(cnt2, cnt)
        assert np.max(np.abs(r2 / ro - 1)) < tol
    assert abs(np.mean(ro) / g.pixarea - 1) < 0.02   # white noise -> pixel area
    c3, r3 = maps.binned_power(m, bin_edges=edges, wcs=g)
    assert np.max(np.abs(r3 / ro - 1)) < 1e-12


def test_matched_filter_coadd_and_template_amplitude():
    """matched_filter, kspace_coadd, MatchedFilter.apply vs the oracle."""
    from orphics_amd import maps
    shape = (128, 128)
    g = geom(shape)
    ml = g.modlmap()
    rng = np.random.default_rng(12)
    m = rng.standard_normal(shape)
    cls = 1e3 / (1 + (np.arange(8000) / 300.) ** 3)
    got = maps.matched_filter(m, 1.5, cls=cls, noise_uk_arcmin=10.0, wcs=g)
    ref = mo.matched_filter(m, 1.5, g.step_y, g.step_x, cls, noise_uk_arcmin=10.0)
    assert rel(got, ref) < 1e-12
    kmaps = np.array([np.fft.fft2(rng.standard_normal(shape)) for _ in range(3)])
    kbeams = np.array([maps.gauss_beam(ml, f) for f in (1.4, 2.2, 5.0)])
    kncovs = np.array([np.full(shape, v) for v in (1e-5, 3e-5, 0.0)])   # a zero-noise map -> non-finite -> 0 rule
    kncovs[2][ml > 2000] = 2e-5
    got = maps.kspace_coadd(kmaps, kbeams, kncovs, fkbeam=kbeams[0])
    ref = mo.kspace_coadd(kmaps, kbeams, kncovs, fkbeam=kbeams[0])
    assert rel(got, ref) < 1e-12
    templ = np.exp(-0.5 * (g.modlmap() * 0 + np.hypot(*np.meshgrid(np.arange(128) - 64, np.arange(128) - 64))) ** 2 / 25.)
    n2d = 1e-4 * (1 + (ml / 1000.) ** 2)
    data = 3.7 * templ + 0.01 * rng.standard_normal(shape)
    mf = maps.MatchedFilter(shape, g, template=templ, noise_power=n2d)
    amp, var = mf.apply(imap=data)
    ramp, rvar = mo.matched_filter_apply(np.fft.fft2(templ), np.fft.fft2(data), n2d, g.area / (128 * 128) ** 2)
    assert abs(amp / ramp - 1) < 1e-12 and abs(var / rvar - 1) < 1e-12
    assert abs(amp - 3.7) < 0.05


def test_split_calc_and_noise_from_splits():
    """split-based power (maps.py:2296-2411) vs the same algebra on the NumPy oracle."""
    from orphics_amd import maps
    shape = (64, 128)
    g = geom(shape)
    fo = mo.FourierCalc(shape, g.step_y, g.step_x)
    rng = np.random.default_rng(30)
    sig = rng.standard_normal(shape)
    splits = np.array([sig + 0.5 * rng.standard_normal(shape) for _ in range(4)])
    ks = np.fft.fft2(splits)
    co = ks.mean(0)
    for alt in (True, False):
        tot, cr, no = maps.split_calc(ks, ks, co, co, alt=alt, wcs=g)
        rt = fo.f2power(co, co)
        if alt:
            rn = sum(fo.f2power(ks[i] - co, ks[i] - co) for i in range(4)) / ((1 - 1. / 4) * 16)
            rc = rt - rn
        else:
            rc = sum(fo.f2power(ks[i], ks[j]) for i in range(4) for j in range(4) if i != j) / 12.
            rn = rt - rc
        assert rel(tot, rt) < 1e-12 and rel(cr, rc) < 1e-11 and rel(no, rn) < 1e-10
    # device-native (HalfPlane) splits give the same binned spectra
    fh = maps.FourierCalc(shape, g, layout="half")
    hs = [fh.fft(torch.as_tensor(s).cuda()) for s in splits]
    hco = type(hs[0])(sum(h.t for h in hs) / 4., hs[0].eng)
    tot_h, cr_h, no_h = maps.split_calc(hs, hs, hco, hco, fourier_calc=fh, alt=True)
    rn = sum(fo.f2power(ks[i] - co, ks[i] - co) for i in range(4)) / ((1 - 1. / 4) * 16)
    assert rel(no_h.numpy(), rn) < 1e-10
    noise, cross_teb = maps.noise_from_splits(splits, wcs=g)
    s32 = splits.astype(np.float32).astype(np.float64)
    k32 = np.fft.fft2(s32)
    auto = sum(fo.f2power(k, k) for k in k32) / 4
    cross = sum(fo.f2power(k32[i], k32[j]) for i in range(4) for j in range(i + 1, 4)) / 6.
    assert rel(noise, (auto - cross) / 4) < 2e-5          # the reference casts splits to float32 (maps.py:2354)
    assert rel(cross_teb, cross) < 2e-5
    # I,Q,U splits: (ncomp, ncomp) matrices; every cross term takes its FIRST component from the earlier split
    # (maps.py:2392-2395 with power2d's assembly, maps.py:1661-1670) -- the running-sum evaluation keeps that order
    iqu = np.array([[sig * (c + 1) + 0.5 * rng.standard_normal(shape) for c in range(3)] for _ in range(3)])
    noise3, cross3 = maps.noise_from_splits(iqu, wcs=g)
    k3 = np.fft.fft2(iqu.astype(np.float32).astype(np.float64))
    assert np.asarray(noise3).shape == (3, 3) + shape and np.asarray(cross3).shape == (3, 3) + shape
    for a in range(3):
        for b in range(a, 3):
            au = sum(fo.f2power(k3[i, a], k3[i, b]) for i in range(3)) / 3.
            cr = sum(fo.f2power(k3[i, a], k3[j, b]) for i in range(3) for j in range(i + 1, 3)) / 3.
            assert rel(np.asarray(noise3)[a, b], (au - cr) / 3.) < 3e-5 and rel(np.asarray(cross3)[a, b], cr) < 3e-5
            assert np.array_equal(np.asarray(cross3)[b, a], np.asarray(cross3)[a, b])


def test_coadd_and_kappa_to_phi_match_the_reference_functions():
    """The product's kspace_coadd and FlatLenser.kappa_to_phi vs outputs of the REFERENCE's own kspace_coadd
    (maps.py:1098-1114) and fkappa_to_fphi (lensing.py:662-665), tests/golden/maps_host_reference.npz (32 x 36 planes:
    zero-noise and zero-beam modes; modes below l = 2)."""
    from orphics_amd import maps, lensing
    gd = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "maps_host_reference.npz"))
    got = maps.kspace_coadd(gd["coadd_kmaps"], gd["coadd_kbeams"], gd["coadd_kncovs"], fkbeam=0.8)
    want = gd["coadd_out"]
    assert np.array_equal(got == 0, want == 0)            # the non-finite -> 0 modes are the same modes
    assert rel(got, want) < 1e-13
    # FourierCalc.power2d's (ncomp, ncomp, Ny, Nx) assembly from supplied transforms (maps.py:1661-1670)
    ka, kb = gd["p2d_k1"], gd["p2d_k2"]
    fc = maps.FourierCalc(ka.shape, geom(ka.shape[-2:]))
    fc.normfact = float(gd["f2_norm"])
    for kw, key in ((dict(kmap2=kb), "p2d_cross"), (dict(), "p2d_auto"), (dict(kmap2=kb, skip_cross=True, pixel_units=True), "p2d_skip_cross_pixel_units")):
        got = np.asarray(fc.power2d(kmap=ka, **kw)[0])
        assert got.shape == gd[key].shape and rel(got, gd[key]) < 1e-13
        assert np.array_equal(got == 0, gd[key] == 0)      # skip_cross leaves the off-diagonal blocks zero
    # kappa -> phi on a real map: the reference function applied to the oracle's DFT of kappa, inverted on the host
    shape = (64, 128)
    g = geom(shape)
    rng = np.random.default_rng(3)
    kappa = rng.standard_normal(shape)
    ml = g.modlmap()
    from oracle import qe_oracle as qo
    phi_ref = np.fft.ifft2(qo.fkappa_to_fphi(np.fft.fft2(kappa), ml)).real
    fl = lensing.FlatLenser(shape, g, dtype="f64")
    phi = fl.kappa_to_phi(kappa)
    phi = phi.cpu().numpy() if hasattr(phi, "cpu") else np.asarray(phi)
    assert rel(phi, phi_ref) < 1e-12
