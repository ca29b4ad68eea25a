"""CPU: the NumPy oracle against the golden vectors generated from the real
reference (tests/golden/make_golden.py) and against analytic known answers."""
import os

import numpy as np
import pytest

from oracle import maps_oracle as mo
from oracle import qe_oracle as qo
from oracle import stats_oracle as so


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "bin2d_reference.npz"))


def test_bin2d_indices_counts_bit_exact(g):
    b = so.bin2D(g["a_modlmap"], g["a_edges"])
    assert np.array_equal(b.digitized, g["a_digitized"])
    c, r, cnt = b.bin(g["a_data"], get_count=True)
    assert np.array_equal(cnt, g["a_count"])
    assert np.array_equal(c, g["a_cents"])
    np.testing.assert_allclose(r, g["a_res"], rtol=1e-15, atol=0)


def test_bin2d_weighted_nan_err(g):
    b = so.bin2D(g["a_modlmap"], g["a_edges"])
    _, r = b.bin(g["a_data"], weights=g["a_weights"])
    np.testing.assert_allclose(r, g["a_res_w"], rtol=1e-15)
    _, r, cnt = b.bin(g["a_data_nan"], mask_nan=True, get_count=True)
    assert np.array_equal(cnt, g["a_count_nan"])
    np.testing.assert_allclose(r, g["a_res_nan"], rtol=1e-15)
    with np.errstate(all="ignore"):
        _, _, s = b.err_reference_shifted(g["a_data"])
    np.testing.assert_allclose(s, g["a_err_ref_shifted"], rtol=1e-13)
    # intended statistic: constant-within-bin data has zero scatter
    const = b.digitized.reshape(g["a_data"].shape).astype(float)
    _, _, s0 = b.bin(const, err=True)
    assert np.allclose(s0, 0)


def test_bin2d_exact_ties_and_quirks(g):
    bt = so.bin2D(g["t_modlmap"], g["t_edges"])
    assert np.array_equal(bt.digitized, g["t_digitized"])
    _, r, cnt = bt.bin(g["t_data"], get_count=True)
    assert np.array_equal(cnt, g["t_count"])
    np.testing.assert_allclose(r, g["t_res"], rtol=1e-15)
    b3 = so.bin2D(g["h3_modrmap"], g["h3_edges"])
    c3, r3 = b3.bin(g["h3_data"])
    assert c3.size == 3 and r3.size == g["h3_res"].size == 2  # H3: a real bin is dropped
    np.testing.assert_allclose(r3, g["h3_res"])
    assert np.array_equal(so.bin2D(g["tie_vals"], g["tie_edges"]).digitized, g["tie_digitized"])
    assert list(g["tie_digitized"]) == [0, 1, 2, 3, 4]


def test_get_stats_and_moments(golden_dir):
    s = np.load(os.path.join(golden_dir, "stats_reference.npz"))
    X = s["X"]
    st = so.get_stats(X)
    for k in ("mean", "cov", "covmean", "err", "errmean", "corr"):
        np.testing.assert_allclose(st[k], s["gs_" + k], rtol=1e-13, atol=1e-15)
    parts = [(10, X[:10].sum(0), X[:10].T @ X[:10]), (27, X[10:].sum(0), X[10:].T @ X[10:])]
    n, S, C = so.moments_merge(parts)
    mean, cov = so.moments_mean_cov(n, S, C)
    np.testing.assert_allclose(mean, s["st_mean"], rtol=1e-13)
    np.testing.assert_allclose(cov, s["st_cov"], rtol=1e-10, atol=1e-12)


def test_mpi_distribute(golden_dir):
    m = np.load(os.path.join(golden_dir, "mpi_reference.npz"))
    for nt, nc in m["pairs"]:
        num_each, dist = so.mpi_distribute(int(nt), int(nc))
        assert np.array_equal(num_each, m[f"num_each_{nt}_{nc}"])
        assert np.array_equal([d[0] for d in dist], m[f"first_{nt}_{nc}"])


# ---- analytic known answers for the unpinned (pixell-boundary) pieces --------
RES = 2.0 * np.pi / 180. / 60.


def test_white_noise_power_is_pixel_area():
    shape = (256, 256)
    fc = mo.FourierCalc(shape, RES, -RES)
    rng = np.random.default_rng(0)
    p2d, _, _ = fc.power2d(rng.standard_normal(shape))
    assert abs(p2d.mean() / RES ** 2 - 1) < 0.02


def test_single_mode_delta_and_unit_filter():
    shape = (64, 64)
    fc = mo.FourierCalc(shape, RES, -RES)
    y, x = np.mgrid[:64, :64]
    m = np.cos(2 * np.pi * (3 * y + 5 * x) / 64.)
    p2d, _, _ = fc.power2d(m)
    nz = np.argwhere(p2d > 1e-12 * p2d.max())
    assert {tuple(v) for v in nz} == {(3, 5), (61, 59)}
    rng = np.random.default_rng(1)
    z = rng.standard_normal(shape)
    assert np.allclose(mo.filter_map(z, np.ones(shape)), z)
    assert mo.mask_kspace(shape, RES, -RES, lmin=300, lmax=2000).dtype.kind == "i"


def test_mapgen_power_matches_input():
    shape = (128, 128)
    ml = mo.modlmap(shape, RES, -RES)
    cov = (1.0 / (1 + (ml / 500.) ** 2)).reshape((1, 1) + shape)
    mg = mo.MapGen(shape, RES, -RES, cov)
    fc = mo.FourierCalc(shape, RES, -RES)
    acc = 0
    for s in range(20):
        m = mg.get_map(seed=s, scalar=True)
        acc = acc + fc.power2d(m)[0]
    ratio = (acc / 20)[ml > 0] / cov[0, 0][ml > 0]
    assert abs(ratio.mean() - 1) < 0.02


def _tt_setup(N=32, res_arcmin=4.0):
    res = res_arcmin * np.pi / 180. / 60.
    shape = (N, N)
    ml = mo.modlmap(shape, res, -res)
    cl = 1e3 / (1 + (ml / 300.) ** 3)
    noise = np.full(shape, (10.0 * np.pi / 180. / 60.) ** 2)
    beam = mo.gauss_beam(ml, 3.0)
    kmask = mo.mask_kspace(shape, res, -res, lmin=100, lmax=1800)
    return shape, res, ml, cl, noise, beam, kmask


def test_qe_response_matches_brute_force():
    shape, res, ml, cl, noise, beam, kmask = _tt_setup()
    q = qo.QEOracleTT(shape, res, -res, cl, cl, noise, beam, kmask)
    ct = cl + noise / beam ** 2
    wg = cl / ct * kmask
    wh = 1. / ct * kmask
    ly, lx = mo.laxes(shape, res, -res)
    for (yi, xi) in [(1, 0), (0, 2), (3, 29), (5, 5)]:
        bf = qo.brute_force_response_tt(ly, lx, q.area, wg, wh, cl, yi, xi)
        assert abs(q.R[yi, xi] / bf - 1) < 1e-10


def test_qe_linear_response_recovers_injected_phi():
    """T' = T + grad(phi).grad(T) to first order: <kappa_hat> must equal the
    injected kappa mode (average over CMB realisations)."""
    shape, res, ml, cl, noise, beam, kmask = _tt_setup(N=64, res_arcmin=3.0)
    nonoise = noise * 0
    q = qo.QEOracleTT(shape, res, -res, cl, cl, nonoise, np.ones(shape), kmask)
    ly, lx = mo.laxes(shape, res, -res)
    LY, LX = np.meshgrid(ly, lx, indexing="ij")
    npix = shape[0] * shape[1]
    phik = np.zeros(shape, complex)
    yi, xi = 2, 3
    amp = 1e-7 * npix
    phik[yi, xi] = amp
    phik[-yi, -xi] = amp
    gpx = np.fft.ifft2(1j * LX * phik).real
    gpy = np.fft.ifft2(1j * LY * phik).real
    rng = np.random.default_rng(3)
    acc = 0
    nsim = 60
    for i in range(nsim):
        tk = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) * np.sqrt(cl * npix / q.pixarea / 2.)
        t = np.fft.ifft2(tk).real * np.sqrt(2.)
        tkk = np.fft.fft2(t)
        tx = np.fft.ifft2(1j * LX * tkk).real
        ty = np.fft.ifft2(1j * LY * tkk).real
        tl = t + gpx * tx + gpy * ty
        # difference with the unlensed reconstruction removes the Gaussian (N0) noise exactly
        acc = acc + (q.kappa_ft(np.fft.fft2(tl)) - q.kappa_ft(tkk))[yi, xi]
    est = acc / nsim
    L = ml[yi, xi]
    expected = L * (L + 1) / 2. * amp
    assert abs(est.real / expected - 1) < 0.05
    assert abs(est.imag / expected) < 0.05


def _pol_setup(N=48, res_arcmin=3.0):
    res = res_arcmin * np.pi / 180. / 60.
    shape = (N, N)
    ml = mo.modlmap(shape, res, -res)
    tt = 1e3 / (1 + (ml / 300.) ** 3)
    cl = dict(TT=tt, EE=0.05 * tt, BB=0.0 * tt, TE=0.12 * tt * np.cos(ml / 400.))
    w = (2.0 * np.pi / 180. / 60.) ** 2
    noise = dict(T=np.full(shape, w), P=np.full(shape, 2 * w))   # filters only (maps are noise-free)
    mk = mo.mask_kspace(shape, res, -res, lmin=100, lmax=1800)
    return shape, res, ml, cl, noise, mk


def test_general_estimators_response_matches_brute_force():
    shape, res, ml, cl, noise, mk = _pol_setup(N=24, res_arcmin=5.0)
    cl = dict(cl, BB=0.002 * cl["TT"])
    q = qo.QEOracle(shape, res, -res, cl, noise, np.ones(shape), dict(T=mk, P=mk))
    for XY in ("TT", "EE", "EB", "TE", "TB"):
        g, f = qo.estimator_terms(XY)
        q.setup(XY)
        for (yi, xi) in [(1, 2), (3, 22)]:
            bf = qo.brute_force_gf(q, g, f, yi, xi)
            assert abs(q.R[XY][yi, xi] / bf - 1) < 1e-10


@pytest.mark.parametrize("XY", ["TT", "EE", "EB", "TE", "TB"])
def test_general_estimators_recover_injected_phi(XY):
    """First-order lensed T,Q,U (scalar remapping) -> E,B by the reference rotation
    (maps.py:1614-1615) -> every estimator returns the injected kappa mode."""
    shape, res, ml, cl, noise, mk = _pol_setup()
    q = qo.QEOracle(shape, res, -res, cl, noise, np.ones(shape), dict(T=mk, P=mk))
    q.setup(XY)
    ly, lx = mo.laxes(shape, res, -res)
    LY, LX = np.meshgrid(ly, lx, indexing="ij")
    npix = shape[0] * shape[1]
    rot = mo.queb_rotmat(mo.lmap(shape, res, -res))
    irot = mo.queb_rotmat(mo.lmap(shape, res, -res), inverse=True)
    phik = np.zeros(shape, complex)
    yi, xi = 2, 3
    amp = 1e-7 * npix
    phik[yi, xi] = amp
    phik[-yi, -xi] = amp
    gpx = np.fft.ifft2(1j * LX * phik).real
    gpy = np.fft.ifft2(1j * LY * phik).real
    rng = np.random.default_rng(5)
    # correlated T,E Gaussian modes; B = 0
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.nan_to_num(cl["TE"] / np.sqrt(cl["TT"] * cl["EE"]))

    def lens(m):
        k = np.fft.fft2(m)
        return m + gpx * np.fft.ifft2(1j * LX * k).real + gpy * np.fft.ifft2(1j * LY * k).real

    acc = 0
    nsim = 40
    for i in range(nsim):
        w1 = np.fft.fft2(rng.standard_normal(shape))
        w2 = np.fft.fft2(rng.standard_normal(shape))
        sc = np.sqrt(1.0 / q.pixarea)
        kT = w1 * np.sqrt(cl["TT"]) * sc
        kE = (r * w1 + np.sqrt(1 - r ** 2) * w2) * np.sqrt(cl["EE"]) * sc
        kB = 0 * kE
        qu = mo.map_mul(irot, np.array([kE, kB]))
        T, Q, U = np.fft.ifft2(kT).real, np.fft.ifft2(qu[0]).real, np.fft.ifft2(qu[1]).real
        f0 = {"T": np.fft.fft2(T)}
        eb = mo.map_mul(rot, np.array([np.fft.fft2(Q), np.fft.fft2(U)]))
        f0["E"], f0["B"] = eb[0], eb[1]
        f1 = {"T": np.fft.fft2(lens(T))}
        eb = mo.map_mul(rot, np.array([np.fft.fft2(lens(Q)), np.fft.fft2(lens(U))]))
        f1["E"], f1["B"] = eb[0], eb[1]
        X, Y = XY[0], XY[1]
        acc = acc + (q.kappa_ft(XY, f1[X], f1[Y]) - q.kappa_ft(XY, f0[X], f0[Y]))[yi, xi]
    est = acc / nsim
    L = ml[yi, xi]
    expected = L * (L + 1) / 2. * amp
    assert abs(est.real / expected - 1) < 0.08, (XY, est / expected)
    assert abs(est.imag / expected) < 0.08


def test_oracle_estimators_unbiased_on_fully_lensed_sims_incl_EB():
    """Independent of the separable-term tables' own consistency checks: NumPy GRFs (E only, B = 0), remapped by the
    full (order-5 Taylor) flat-sky lensing operation with a Gaussian kappa, beam + white noise, E/B by the reference
    rotation -> the TT, EE and EB estimators' cross-power with the input kappa equals the input auto-power.  (The
    first-order injected-phi test above cannot see a wrong EB normalisation at second order or a beam/noise slip.)"""
    from orphics_amd import cosmology
    N, res = 128, 2.0 * np.pi / 180. / 60.
    shape = (N, N)
    th = cosmology.default_theory()
    ml = mo.modlmap(shape, res, -res)
    lm = mo.lmap(shape, res, -res)
    rot, irot = mo.queb_rotmat(lm), mo.queb_rotmat(lm, inverse=True)
    pix = mo.planar_area(shape, res, -res) / (N * N)
    rng = np.random.default_rng(12)

    def grf(c2d):
        return np.fft.ifft2(np.fft.fft2(rng.standard_normal(shape)) * np.sqrt(np.maximum(c2d, 0) / pix)).real
    beam = mo.gauss_beam(ml, 1.5)
    nT = mo.white_noise_power(1.0)
    cl = {k: th.lCl(k, ml) for k in ("TT", "EE", "BB", "TE")}
    mask = ((ml > 300) & (ml < 2500)).astype(float)
    q = qo.QEOracle(shape, res, -res, cl, {"T": np.full(shape, nT), "P": np.full(shape, 2 * nT)}, beam, {"T": mask, "P": mask})
    sel = (ml > 100) & (ml < 1500)
    acc = {"TT": [], "EE": [], "EB": []}
    for _ in range(10):
        T, E, kap = grf(th.uCl("TT", ml)), grf(th.uCl("EE", ml)), grf(th.gCl("kk", ml))
        kE = np.fft.fft2(E)
        Q, U = np.fft.ifft2(irot[0, 0] * kE).real, np.fft.ifft2(irot[1, 0] * kE).real
        alpha = qo.alpha_from_kappa(kap, res, -res)
        obs = [np.fft.ifft2(np.fft.fft2(qo.flat_taylens(alpha, m, res, -res, taylor_order=5)) * beam).real + grf(np.full(shape, n))
               for m, n in zip((T, Q, U), (nT, 2 * nT, 2 * nT))]
        kT, kQ, kU = [np.fft.fft2(m) for m in obs]
        f = {"T": kT, "E": rot[0, 0] * kQ + rot[0, 1] * kU, "B": rot[1, 0] * kQ + rot[1, 1] * kU}
        kk = np.fft.fft2(kap)
        for est in acc:
            rec = q.kappa_ft(est, f[est[0]], f[est[1]])
            acc[est].append((rec * np.conj(kk)).real[sel].sum() / (np.abs(kk) ** 2)[sel].sum())
    for est, v in acc.items():
        mean, sem = np.mean(v), np.std(v) / np.sqrt(len(v))
        assert abs(mean - 1) < 0.05 + 3 * sem, (est, mean, sem)


def test_maps_host_helpers_match_the_reference_functions():
    """gauss_beam, cosine_window and FourierCalc.f2power's body: outputs of the REFERENCE's own function definitions
    (executed out of maps.py by tests/golden/make_golden_maps_host.py) vs the oracle and vs the product's host helpers."""
    import os
    from oracle import maps_oracle as mo
    from orphics_amd import maps
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "maps_host_reference.npz"))
    for i in range(2):
        fw = float(g["beam_fwhm_%d" % i])
        assert np.array_equal(mo.gauss_beam(g["beam_ell"], fw), g["beam_out_%d" % i])
        assert np.array_equal(maps.gauss_beam(g["beam_ell"], fw), g["beam_out_%d" % i])
    for i, (Ny, Nx, ay, ax, py, px) in enumerate(g["win_cases"]):
        want = g["win_out_%d" % i]
        assert np.array_equal(mo.cosine_window(int(Ny), int(Nx), int(ay), int(ax), int(py), int(px)), want)
        assert np.array_equal(maps.cosine_window(int(Ny), int(Nx), lenApodY=int(ay), lenApodX=int(ax), padY=int(py), padX=int(px)), want)
    k1, k2, norm = g["f2_k1"], g["f2_k2"], float(g["f2_norm"])
    fc = mo.FourierCalc(k1.shape, 1e-3, -1e-3)
    fc.normfact = norm
    assert np.array_equal(fc.f2power(k1, k2), g["f2_out"])
    assert np.array_equal(fc.f2power(k1, k2, pixel_units=True), g["f2_out_pixel_units"])
    # power2d's multi-component assembly (maps.py:1661-1670): autos, mirrored upper-triangle crosses, skip_cross, pixel_units
    ka, kb = g["p2d_k1"], g["p2d_k2"]
    assert np.array_equal(fc.power2d(kmap=ka, kmap2=kb)[0], g["p2d_cross"])
    assert np.array_equal(fc.power2d(kmap=ka)[0], g["p2d_auto"])
    assert np.array_equal(fc.power2d(kmap=ka, kmap2=kb, skip_cross=True, pixel_units=True)[0], g["p2d_skip_cross_pixel_units"])
    # kspace_coadd (maps.py:1098-1114) with zero-noise / zero-beam modes, and lensing.fkappa_to_fphi (lensing.py:662-665)
    from oracle import qe_oracle as qo
    assert np.array_equal(mo.kspace_coadd(g["coadd_kmaps"], g["coadd_kbeams"], g["coadd_kncovs"], fkbeam=0.8), g["coadd_out"])
    assert np.array_equal(qo.fkappa_to_fphi(g["fphi_fkappa"], g["fphi_modlmap"]), g["fphi_out"])
    assert np.all(g["fphi_out"][g["fphi_modlmap"] < 2.] == 0)


def _bilinear_qfrag(U, V):
    """The data-defined two-leg map of tests/golden/make_golden_splits.py."""
    return lambda a, b: U * a * b + V * a * np.roll(b, (1, 2), (0, 1))


def test_split_estimators_match_the_reference_functions():
    """SplitLensing.cross_estimator (lensing.py:980-1003) and split_calc (maps.py:2296-2333): outputs of the REFERENCE's
    own definitions (tests/golden/make_golden_splits.py) vs the oracle restatements."""
    import os
    from oracle import maps_oracle as mo
    from oracle import qe_oracle as qo
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "splits_reference.npz"))
    norm = float(g["normfact"])
    power = lambda a, b: np.real(np.conj(a) * b) * norm       # noqa: E731  (f2power, pinned above)
    qfrag = _bilinear_qfrag(g["U"], g["V"])
    for n in (4, 5, 6):
        got = qo.split_cross_estimator(qfrag, power, g["cross_splits_%d" % n])
        want = g["cross_out_%d" % n]
        assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
    isp, jsp = g["sc_isplits"], g["sc_jsplits"]
    for alt, tag in ((True, "alt"), (False, "loop")):
        t, c, nz = mo.split_calc(isp, jsp, isp.mean(0), jsp.mean(0), power, alt=alt)
        for got, key in ((t, "sc_total_"), (c, "sc_crosses_"), (nz, "sc_noise_")):
            want = g[key + tag]
            assert np.abs(got - want).max() <= 1e-13 * np.abs(want).max(), (tag, key)
