"""GPU, BASELINE.json full sizes: size-independent properties (round trips, Parseval, linearity, checksums of
the bit-exact bin ids, f32-vs-f64 agreement) where the NumPy oracle would take minutes."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def eng(n, prec):
    from orphics_amd.engine import Engine
    return Engine.get(n, n, prec)


@pytest.mark.parametrize("N", [4096, 8192, 16384])
def test_fft_round_trip_parseval_linearity(N):
    e = eng(N, "f32")
    x = e.randn(11, 0)
    y = e.randn(11, 1)
    kx = e.rfft(x)
    back = e.irfft(kx)
    err = (back - x).abs().max().item()
    assert err < 2e-5, err                                   # encode -> decode
    # Parseval with Hermitian multiplicities: sum |x|^2 = (1/Npix) sum_full |X|^2
    w = kx[:, :N // 2 + 1].abs().double() ** 2
    tot = 2 * w.sum() - w[:, 0].sum() - w[:, N // 2].sum()
    assert abs(tot.item() / (N * N) / (x.double() ** 2).sum().item() - 1) < 1e-5
    # linearity: F(2x - 3y) = 2F(x) - 3F(y)
    ky = e.rfft(y)
    kz = e.rfft(e.axpby(x, y, 2.0, -3.0))
    ref = 2.0 * kx - 3.0 * ky
    assert ((kz - ref)[:, :N // 2 + 1].abs().max() / ref[:, :N // 2 + 1].abs().max()).item() < 2e-6
    # a single Fourier mode comes back as a delta
    del kx, ky, kz, ref, back
    k1 = e.hc()
    k1[5, 7] = N * N / 2.0
    m = e.irfft(k1)
    yy = torch.arange(N, device=m.device, dtype=torch.float64)
    expect = torch.cos(2 * np.pi * (5 * yy[:, None] + 7 * yy[None, :]) / N)
    assert (m.double() - expect).abs().max().item() < 2e-4


@pytest.mark.parametrize("N,res", [(4096, 0.5), (8192, 0.5), (16384, 0.25)])
def test_bin_ids_bit_exact_with_numpy_at_full_size(N, res):
    """H2: Delta ell = 21600/4096 makes grid modes land exactly on integer edges; device ids must equal
    np.digitize(right=True) of NumPy's modlmap everywhere, and counts must sum to Npix.  16384^2 0.25' (BASELINE config 5)
    has the same Delta ell = 5.2734375 tie structure as 8192^2 0.5'."""
    from orphics_amd.geometry import FlatGeometry
    e = eng(N, "f32")
    g = FlatGeometry.from_res((N, N), res)
    ly, lx = g.laxes()
    e.set_laxes(ly, lx)
    edges = np.arange(0., 12000., 5400. / 8)                 # integer edges hit exactly by on-axis modes
    ed = torch.as_tensor(edges, device=e.device)
    ids_h = e.modl_digitize(ed, half=True).cpu().numpy()[:, :N // 2 + 1]
    ml_h = np.sqrt(ly[:, None] ** 2 + lx[None, :N // 2 + 1] ** 2)
    ref = np.digitize(ml_h.reshape(-1), edges, right=True).reshape(ml_h.shape)
    assert np.array_equal(ids_h, ref)
    ties = np.isin(ml_h, edges).sum()
    assert ties > 10                                        # the exact-tie cases are really exercised
    data = torch.ones((N, e.kp), dtype=torch.float32, device=e.device)
    sums, counts = e.bin(data, e.modl_digitize(ed, half=True), len(edges) + 1, herm=True)
    assert int(counts.sum().item()) == N * N
    assert torch.equal(counts.double(), sums)


def test_config2_tt_qe_4096_f32_vs_f64():
    """BASELINE config 2: 4096^2 0.5' TT QE, one realisation: f32 kernels vs f64 kernels (the f64 path is the one
    pinned to the NumPy oracle at small sizes): kappa bandpowers within 1e-5, fused == modular."""
    from orphics_amd import cosmology, lensing, maps
    from orphics_amd.geometry import FlatGeometry
    N, res = 4096, 0.5
    shape = (N, N)
    g = FlatGeometry.from_res(shape, res)
    th = cosmology.default_theory()
    ml = g.modlmap()
    beam = maps.gauss_beam(ml, 1.5)
    noise = np.full(shape, cosmology.white_noise_power(1.0))
    tmask = maps.mask_kspace(shape, g, lmin=300, lmax=2000)
    kmask = maps.mask_kspace(shape, g, lmin=20, lmax=3500)
    out = {}
    for prec in ("f32", "f64"):
        q = lensing.qest(shape, g, th, noise2d=noise, beam2d=beam, kmask=tmask, kmask_K=kmask,
                         unlensed_equals_lensed=True, dtype=prec)
        e = q.eng
        t = e.randn(5, 0) if prec == "f32" else out["t"].double()
        out["t"] = t if prec == "f32" else out["t"]
        kT = e.rfft(t.contiguous())
        kk = q.reconstruct_tt_hc(kT).clone()
        km = q.reconstruct_tt_hc(kT, fused=False)
        assert ((kk - km)[:, :N // 2 + 1].abs().max() / km[:, :N // 2 + 1].abs().max()).item() < (3e-5 if prec == "f32" else 1e-11)
        ed = torch.as_tensor(np.linspace(20, 3500, 20), device=e.device)
        s, c = e.bin_power(kk, kk, g.area / float(N * N) ** 2, e.modl_digitize(ed, half=True), 21, herm=True)
        out[prec] = (s[1:-1] / c[1:-1].double()).cpu().numpy()
        del q
    assert np.max(np.abs(out["f32"] / out["f64"] - 1)) < 1e-5


def test_config5_16384_tt_qe_runs_and_matches_knox_scatter():
    """BASELINE config 5 (HBM-capacity stress): 16384^2 0.25' TT QE on one GPU; the Gaussian bandpower
    scatter of N0 realisations agrees with the mode-count Knox variance 2 C_b^2 / N_modes."""
    from orphics_amd import cosmology, lensing, maps, mc, stats
    from orphics_amd.geometry import FlatGeometry
    N, res = 16384, 0.25
    shape = (N, N)
    g = FlatGeometry.from_res(shape, res)
    th = cosmology.default_theory()
    nxh = N // 2
    ly, lx = g.laxes()
    ml_h = np.sqrt(ly[:, None] ** 2 + lx[None, :nxh + 1] ** 2)

    def full(a_h):
        o = np.empty(shape, dtype=a_h.dtype)
        o[:, :nxh + 1] = a_h
        o[:, nxh + 1:] = a_h[(-np.arange(N)) % N][:, 1:nxh][:, ::-1]
        return o
    beam_h = maps.gauss_beam(ml_h, 1.5)
    noise_h = np.full(ml_h.shape, cosmology.white_noise_power(1.0))
    q = lensing.qest(shape, g, th, noise2d=full(noise_h), beam2d=full(beam_h),
                     kmask=full(((ml_h > 300) & (ml_h < 2000)).astype(np.int64)),
                     kmask_K=full(((ml_h > 20) & (ml_h < 3500)).astype(np.int64)), unlensed_equals_lensed=True, dtype="f32")
    tot_h = th.lCl("TT", ml_h) * beam_h ** 2 + noise_h
    edges = np.linspace(200, 3000, 15)
    drv = mc.GaussianN0MonteCarlo(q, tot_h, edges, base_seed=5)
    st = drv.run(12)
    mean, var = st.mean("n0"), st.var("n0")
    e = q.eng
    ed = torch.as_tensor(edges, device=e.device)
    ones = torch.ones((N, e.kp), dtype=torch.float32, device=e.device)
    _, counts = e.bin(ones, e.modl_digitize(ed, half=True), len(edges) + 1, herm=True)
    nmodes = counts[1:-1].cpu().numpy() / 2.0              # independent modes of a real field
    knox = cosmology.knox_cov(mean, nmodes)
    ratio = var / knox
    assert 0.3 < np.median(ratio) < 2.5, ratio             # 12 sims: chi^2_11 scatter on each variance
    nl = q.Nlkk["TT"]
    sel = (ml_h > 400) & (ml_h < 2800)
    assert abs(mean.mean() / nl[sel].mean() - 1) < 0.25    # N0 level


@pytest.mark.parametrize("N,tlmax", [(2048, 2000), (4096, 2000), (8192, 2000), (4096, 6000), (8192, 6000), (8192, 4000), (16384, 2000)])
def test_tt_bandpowers_match_numpy_oracle_at_full_size(N, tlmax):
    """The north-star parity statement checked directly at BASELINE config-2 size: the default (pruned, f32) device
    path -- R2C of the map, fused estimator, |kappa_hat|^2 bandpowers -- against the float64 full-plane NumPy
    oracle on the same map: bin ids bit-exact, bandpowers within 1e-5 relative (f64 kernels: 1e-9).  8192 is the
    size the headline metric is quoted on.  tlmax = 6000: SURVEY 8(d)'s high-resolution variant (T filter ell in (300, 6000):
    569 leg columns at 4096^2, column grid 2048 = ny / 2, row grid 2048) -- the geometry class that runs other kernels than
    the headline's (R = 2); at 8192^2 (1138 leg columns, column grid 4096, row grid 4096) it is the R = 2 split: the wide-band row R2C
    (row_r2c_rs_body<T, 12, 1, .., 5>), col_fband<LR = 1> and the R = 2 layout of the row stage; tlmax = 4000 (759 leg columns) is the same
    path with two live 512-column blocks per side in the row stage instead of four.  16384 (0.25', BASELINE config 5; 45 s of NumPy on 64 host threads): the R = 8 row R2C, col_fband<LR = 3> and the
    R = 8 layout of the row stage against NumPy on a real 16384^2 map (first run: profiles/r05_16384_oracle.txt)."""
    import time
    from orphics_amd import cosmology, lensing, maps
    from orphics_amd.geometry import FlatGeometry
    from oracle import maps_oracle as mo
    from oracle import qe_oracle as qo
    from oracle import stats_oracle as so
    res = 0.25 if N == 16384 else 0.5
    shape = (N, N)
    mo.set_workers(min(64, os.cpu_count() or 1) if N >= 8192 else 1)
    g = FlatGeometry.from_res(shape, res)
    th = cosmology.default_theory()
    ml = g.modlmap()
    beam = maps.gauss_beam(ml, 1.5)
    noise = np.full(shape, cosmology.white_noise_power(1.0))
    tmask = maps.mask_kspace(shape, g, lmin=300, lmax=tlmax)
    kmask = maps.mask_kspace(shape, g, lmin=20, lmax=3500)
    cltt = th.lCl("TT", ml)
    rng = np.random.default_rng(11)
    tk = np.fft.fft2(rng.standard_normal(shape)) * np.sqrt((cltt * beam ** 2 + noise) / g.pixarea)
    tmap = np.fft.ifft2(tk).real
    edges = np.linspace(20, 3500, 20)
    t0 = time.time()
    qr = qo.QEOracleTT(shape, g.step_y, g.step_x, cltt, cltt, noise, beam, tmask, kmask_K=kmask)
    ref = qr.kappa_from_map("TT", tmap)
    bo = so.bin2D(ml, edges)
    _, p1r = bo.bin(mo.FourierCalc(shape, g.step_y, g.step_x).power2d(ref)[0])
    t_oracle = time.time() - t0
    for prec, tol in (("f32", 1e-5), ("f64", 1e-9)):
        q = lensing.qest(shape, g, th, noise2d=noise, beam2d=beam, kmask=tmask, kmask_K=kmask,
                         unlensed_equals_lensed=True, dtype=prec)
        assert q.leg_cols > 0 and q.kappa_rows > 0                      # the pruned path is the one under test
        e = q.eng
        x = torch.as_tensor(tmap, dtype=e.rdt, device=e.device)
        kT = e.rfft(x, width=q.leg_cols, rband=q.leg_rows)
        kk = q.reconstruct_tt_hc(kT)
        ids = e.modl_digitize(torch.as_tensor(edges, device=e.device), half=True)
        _, counts = e.bin_power(kk, kk, 1.0, ids, len(edges) + 1, herm=True)
        sums, _ = e.bin_power(kk, kk, g.area / float(N * N) ** 2, ids, len(edges) + 1, herm=True,
                              active_cols=q.kappa_cols, active_rows=q.kappa_rows)
        # integer side: half-plane ids expanded with the Hermitian multiplicity == the oracle's full-plane counts
        assert np.array_equal(counts[1:-1].cpu().numpy(), np.bincount(bo.digitized, minlength=len(edges) + 1)[1:len(edges)])
        p1d = (sums[1:-1] / counts[1:-1].double()).cpu().numpy()
        err = np.max(np.abs(p1d / p1r - 1))
        assert err < tol, "%s: bandpowers differ from the oracle by %.3g" % (prec, err)
        # THE BENCHMARKED CALL (bench.py Runner.run): bins bound to the plan, two real maps per C-ABI call
        # (oa_qe_tt_moments2: row R2C -> col_fwdlegs_cg -> row_qe_pair -> col_div -> binned power -> moment tail), against the
        # same oracle bandpowers.  Second map = 0.5 x the first, rolled: kappa_hat is bilinear and translation-covariant, so
        # its oracle bandpowers are b / 16 exactly; the two maps occupy different slots of every shared launch.
        d = len(edges) - 1
        x1 = (0.5 * torch.roll(x, shifts=(17, 5), dims=(0, 1))).contiguous()
        n = torch.zeros(1, dtype=torch.int64, device=e.device)
        S = torch.zeros(d, dtype=torch.float64, device=e.device)
        C = torch.zeros(d, d, dtype=torch.float64, device=e.device)
        q.bind_bins(ids, len(edges) + 1, g.area / float(N * N) ** 2)
        assert np.array_equal(q.bin_counts().cpu().numpy(), counts.cpu().numpy())      # plan-side mode counts: bit-exact
        q.tt_moments2(x, x1, n, S, C)
        torch.cuda.synchronize()
        assert int(n.item()) == 2
        errS = np.max(np.abs(S.cpu().numpy() / (p1r * (1 + 1 / 16.)) - 1))
        errC = np.max(np.abs(C.cpu().numpy() / (np.outer(p1r, p1r) * (1 + 1 / 256.)) - 1))
        assert errS < tol and errC < 2 * tol, "%s: moments of the two-map call differ from the oracle: %.3g %.3g" % (prec, errS, errC)
        # ... and one map per call (oa_qe_tt_moments) on the second map alone
        n.zero_(); S.zero_(); C.zero_()
        q.tt_moments(x1, n, S, C)
        torch.cuda.synchronize()
        err1 = np.max(np.abs(S.cpu().numpy() * 16. / p1r - 1))
        assert int(n.item()) == 1 and err1 < tol, "%s: one-map call: %.3g" % (prec, err1)
        del q, x1
    mo.set_workers(1)
    print("N = %d: oracle %.1f s; f32 and f64 bandpowers, two-map and one-map moment entries within 1e-5 / 1e-9 of NumPy" % (N, t_oracle))
    assert t_oracle > 0


def _pol_inputs(N, res, th, beam_h, noise_T, seed=21):
    """device T, E, B transforms (f64 hc planes) of Gaussian fields with the observed total spectra"""
    from orphics_amd.geometry import FlatGeometry
    e = eng(N, "f64")
    g = FlatGeometry.from_res((N, N), res)
    ly, lx = g.laxes()
    ml_h = np.sqrt(ly[:, None] ** 2 + lx[None, :N // 2 + 1] ** 2)
    out = []
    for i, (sp, nz) in enumerate((("TT", noise_T), ("EE", 2 * noise_T), ("BB", 2 * noise_T))):
        tot = th.lCl(sp, ml_h) * beam_h ** 2 + nz
        cs = e.hcreal()
        cs[:, :N // 2 + 1] = torch.as_tensor(np.sqrt(tot * float(N * N) ** 2 / g.area), device=e.device)
        out.append(e.grf_hc(seed, i, cs))
        del cs
    return out


_POL_CACHE = {}


def _pol_qest(N, res=0.5):
    """The five-estimator float64 set-up at (N, N) 0.5' with the reference's band limits -- filters, A_L of TT, TE, EE, EB, TB, MV
    weights: by far the longest part of the two config-3 tests at 8192^2 (minutes of host-side NumPy on 67 M-mode planes), built ONCE
    for both."""
    from orphics_amd import cosmology, lensing, maps
    from orphics_amd.geometry import FlatGeometry
    if N not in _POL_CACHE:
        shape = (N, N)
        g = FlatGeometry.from_res(shape, res)
        th = cosmology.default_theory()
        beam = maps.gauss_beam(g.modlmap(), 1.5)
        nT = cosmology.white_noise_power(1.0)
        noise = np.full(shape, nT)
        tmask = maps.mask_kspace(shape, g, lmin=300, lmax=2000)
        kmask = maps.mask_kspace(shape, g, lmin=20, lmax=3500)
        q64 = lensing.qest(shape, g, th, noise2d=noise, beam2d=beam, kmask=tmask, noise2d_P=2 * noise, kmask_P=tmask, kmask_K=kmask,
                           pol=True, unlensed_equals_lensed=True, dtype="f64")
        _POL_CACHE.clear()                   # (one size at a time: the 8192^2 set-up holds gigabytes of filter planes)
        _POL_CACHE[N] = dict(q=q64, g=g, th=th, beam=beam, nT=nT, noise=noise, tmask=tmask, kmask=kmask)
    return _POL_CACHE[N]


@pytest.mark.parametrize("N", [2048, 8192])
def test_config3_mv_f32_vs_f64_and_fused_vs_modular(N):
    """BASELINE config 3: minimum-variance combination of TT, TE, EE, EB, TB on 0.5' maps (8192^2 = the configured
    size).  f32 kernels vs f64 kernels on the kappa bandpowers (< 1e-5, the north-star tolerance), every single
    estimator's bandpowers likewise, and -- at the smaller size -- the fused one-call path vs the modular chain of
    public calls (3 C2R, 2 products, 2 R2C per piece)."""
    res = 0.5
    c = _pol_qest(N, res)
    g, th, beam, nT, q64 = c["g"], c["th"], c["beam"], c["nT"], c["q"]
    k64 = _pol_inputs(N, res, th, beam[:, :N // 2 + 1], nT)
    edges = np.linspace(20, 3500, 20)
    ests = ("TT", "TE", "EE", "EB", "TB")
    res_p = {}
    for prec in ("f64", "f32"):
        q = q64.astype(prec)             # one set-up (f64 kernels), two sets of per-map kernels
        e = q.eng
        kT, kE, kB = [k.to(e.cdt) for k in k64]
        ids = e.modl_digitize(torch.as_tensor(edges, device=e.device), half=True)
        norm = g.area / float(N * N) ** 2
        _, counts = e.bin_power(kT, kT, norm, ids, len(edges) + 1, herm=True)

        def bp(kk):
            s, _ = e.bin_power(kk, kk, norm, ids, len(edges) + 1, herm=True)
            return (s[1:-1] / counts[1:-1].double()).cpu().numpy()
        f = {"T": kT, "E": kE, "B": kB}
        out = {"MV": bp(q.reconstruct_mv_hc(kT, kE, kB, estimators=ests))}
        for XY in ests:
            out[XY] = bp(q.reconstruct_hc(XY, f[XY[0]], f[XY[1]]))
        if N <= 2048 and prec == "f64":
            acc = e.hc()
            for i, XY in enumerate(ests):
                q._reconstruct_hc_modular(XY, f[XY[0]], f[XY[1]], out=acc, norm=q._mv[1][XY], accumulate=(i > 0))
            mod = bp(acc)
            assert np.max(np.abs(mod / out["MV"] - 1)) < 1e-9, "fused MV differs from the modular chain"
        assert q.Nlkk["MV"].shape == (N, N // 2 + 1)
        res_p[prec] = out
        del kT, kE, kB, f
        if prec == "f32":
            del q
        torch.cuda.empty_cache()
    for key in res_p["f64"]:
        err = np.max(np.abs(res_p["f32"][key] / res_p["f64"][key] - 1))
        assert err < 1e-5, "%s: f32 bandpowers differ from f64 by %.3g" % (key, err)


def test_config4_mc_n0_and_mean_field_1000_sims_4096():
    """BASELINE config 4 on one GPU: 1000 Gaussian realisations at 4096^2 0.5' through mc.GaussianN0MonteCarlo.run
    (one oa_mc_run call for the whole shard, device-resident Statistics).  The Monte-Carlo N0 equals the analytic
    N_L^kk from A_L within the Monte-Carlo error in every bin, and the mean field of Gaussian sims is noise."""
    from orphics_amd import cosmology, lensing, maps, mc, stats
    from orphics_amd.geometry import FlatGeometry
    N, res, nsims = 4096, 0.5, 1000
    shape = (N, N)
    g = FlatGeometry.from_res(shape, res)
    th = cosmology.default_theory()
    ml = g.modlmap()
    beam = maps.gauss_beam(ml, 1.5)
    noise = np.full(shape, cosmology.white_noise_power(1.0))
    tmask = maps.mask_kspace(shape, g, lmin=300, lmax=2000)
    kmask = maps.mask_kspace(shape, g, lmin=20, lmax=3500)
    q = lensing.qest(shape, g, th, noise2d=noise, beam2d=beam, kmask=tmask, kmask_K=kmask, unlensed_equals_lensed=True, dtype="f32")
    tot_h = (th.lCl("TT", ml) * beam ** 2 + noise)[:, :N // 2 + 1]
    edges = np.linspace(20, 3500, 20)
    drv = mc.GaussianN0MonteCarlo(q, tot_h, edges, base_seed=1234, mean_field=True)
    st = drv.run(nsims)
    assert st.count("n0") == nsims and st.stack_count("mf") == nsims
    _, nl = stats.bin2D(ml, edges).bin(q.N_kappa("TT"))
    mean = st.mean("n0")
    sem = np.sqrt(st.var("n0") / nsims)
    pull = (mean - nl) / sem
    assert np.all(np.abs(mean / nl - 1) < 0.01), mean / nl       # sub-per-cent in every bin
    assert np.abs(pull).max() < 4.5, pull                       # and within the Monte-Carlo error
    # mean field: |<kappa_hat>|^2 ~ N0 / nsims (pure noise), stays on the GPU
    mf = st.stack_sum("mf", on_device=True)
    mfk = torch.view_as_complex(mf.contiguous())[:, :N // 2 + 1] / float(nsims)
    p_mf = (mfk.abs() ** 2 * (g.area / float(N * N) ** 2)).cpu().numpy()
    mlh = ml[:, :N // 2 + 1]
    sel = (mlh > 300) & (mlh < 3000)
    ratio = p_mf[sel].mean() / (q.Nlkk["TT"][sel].mean() / nsims)
    assert 0.8 < ratio < 1.2, ratio


def _full_from_half(a_h, N):
    """real even plane on the half grid (N, N/2+1) -> full (N, N)"""
    nxh = N // 2
    o = np.empty((N, N), dtype=a_h.dtype)
    o[:, :nxh + 1] = a_h
    o[:, nxh + 1:] = a_h[(-np.arange(N)) % N][:, 1:nxh][:, ::-1]
    return o


def test_config5_16384_strict_from_map_path_vs_dense_and_f32_vs_f64():
    """BASELINE config 5 geometry (16384^2 0.25'), the strict gate: ONE map through the benchmarked from-map call
    (``tt_moments``: row R2C with the R-split, single-pass column stage, row stage on the coarse grids, divergence + binning)
    in float64 against (i) the DENSE float64 pipeline (prune=False: every column and row, the multi-pass kernels whose
    oracle parity is pinned at 2048..8192^2) on the same map: bandpowers within 1e-9, and (ii) the float32 kernels: within
    1e-5 (the north-star tolerance).  Bin ids at this size are covered bit-exactly by
    test_bin_ids_bit_exact_with_numpy_at_full_size."""
    from orphics_amd import cosmology, lensing, maps
    from orphics_amd.geometry import FlatGeometry
    N, res = 16384, 0.25
    shape = (N, N)
    g = FlatGeometry.from_res(shape, res)
    th = cosmology.default_theory()
    nxh = N // 2
    ly, lx = g.laxes()
    ml_h = np.sqrt(ly[:, None] ** 2 + lx[None, :nxh + 1] ** 2)
    beam = _full_from_half(maps.gauss_beam(ml_h, 1.5), N)
    noise = np.full(shape, cosmology.white_noise_power(1.0))
    tmask = _full_from_half(((ml_h > 300) & (ml_h < 2000)).astype(np.int64), N)
    kmask = _full_from_half(((ml_h > 20) & (ml_h < 3500)).astype(np.int64), N)
    del ml_h
    kw = dict(noise2d=noise, beam2d=beam, kmask=tmask, kmask_K=kmask, unlensed_equals_lensed=True)
    edges = np.linspace(20, 3500, 20)
    d = len(edges) - 1
    q64 = lensing.qest(shape, g, th, dtype="f64", **kw)
    e = q64.eng
    x = e.randn(7, 0)
    ids = e.modl_digitize(torch.as_tensor(edges, device=e.device), half=True)
    norm = g.area / float(N * N) ** 2
    out = {}
    for prec in ("f64", "f32"):
        q = q64 if prec == "f64" else q64.astype("f32")
        ee = q.eng
        xx = x if prec == "f64" else x.float()
        n = torch.zeros(1, dtype=torch.int64, device=ee.device)
        S = torch.zeros(d, dtype=torch.float64, device=ee.device)
        C = torch.zeros(d, d, dtype=torch.float64, device=ee.device)
        q.bind_bins(ids, len(edges) + 1, norm)
        q.tt_moments(xx, n, S, C)
        torch.cuda.synchronize()
        assert int(n.item()) == 1
        out[prec] = S.cpu().numpy().copy()
        # the kappa-producing entry (oa_qe_tt from the map) binned by the public histogram call: same bandpowers
        kk = q.reconstruct_tt_from_map(xx)
        s2, c2 = ee.bin_power(kk, kk, norm, ids, len(edges) + 1, herm=True)
        np.testing.assert_allclose((s2[1:-1] / c2[1:-1].double()).cpu().numpy(), out[prec], rtol=(1e-12 if prec == "f64" else 2e-6))
        del kk, xx
        if prec == "f32":
            del q
        torch.cuda.empty_cache()
    assert np.all(out["f64"] > 0)
    err32 = np.max(np.abs(out["f32"] / out["f64"] - 1))
    assert err32 < 1e-5, "16384^2: f32 bandpowers differ from f64 by %.3g" % err32
    del q64
    torch.cuda.empty_cache()
    qd = lensing.qest(shape, g, th, dtype="f64", prune=False, **kw)
    ed = qd.eng
    kk = qd.reconstruct_tt_hc(ed.rfft(x))
    s, c = ed.bin_power(kk, kk, norm, ids, len(edges) + 1, herm=True)
    dense = (s[1:-1] / c[1:-1].double()).cpu().numpy()
    err = np.max(np.abs(out["f64"] / dense - 1))
    assert err < 1e-9, "16384^2: pruned from-map path differs from the dense pipeline by %.3g" % err


def test_config3_mv_8192_f64_matches_numpy_oracle_per_map_arithmetic():
    """BASELINE config 3 at its configured size, against NumPy: the five-estimator MV reconstruction of ONE (T, E, B) set at
    8192^2 0.5' in float64 -- one ``oa_qe_mv`` call: 17 batched leg planes, the estimator-chain row stage on the 2048-point row
    grid / 2048-row column grid (other template instances than the 2048^2 oracle test's), batched divergence -- against
    ``oracle.QEOracle.unnormalized_ft`` (full-plane complex128 NumPy legs, real-space products, divergence) for every
    estimator, combined with the device estimator's per-mode weight x normalisation planes (geometry-generic float64 FFT
    convolutions, pinned against the oracle's A_L / N_L at 128..2048^2 in test_lensing_gpu.py): kappa bandpowers within 1e-9,
    per estimator and for the MV sum."""
    from oracle import maps_oracle as mo
    from oracle import qe_oracle as qo
    from oracle import stats_oracle as so
    N, res = 8192, 0.5
    shape = (N, N)
    c = _pol_qest(N, res)                  # (shared with test_config3_mv_f32_vs_f64_and_fused_vs_modular[8192])
    g, th, beam, tmask, kmask, q = c["g"], c["th"], c["beam"], c["tmask"], c["kmask"], c["q"]
    nT = c["noise"]
    nP = 2 * nT
    ml = g.modlmap()
    ests = ("TT", "TE", "EE", "EB", "TB")
    e = q.eng
    k64 = _pol_inputs(N, res, th, beam[:, :N // 2 + 1], float(nT[0, 0]))
    kmv = q.reconstruct_mv_hc(*k64, estimators=ests).clone()
    kone = {XY: q.reconstruct_hc(XY, k64["TEB".index(XY[0])], k64["TEB".index(XY[1])]).clone() for XY in ("TE", "EB")}
    edges = np.linspace(20, 3500, 20)
    ids = e.modl_digitize(torch.as_tensor(edges, device=e.device), half=True)
    norm = g.area / float(N * N) ** 2

    def bp(kk):
        s, c = e.bin_power(kk, kk, norm, ids, len(edges) + 1, herm=True)
        return (s[1:-1] / c[1:-1].double()).cpu().numpy()
    got = {"MV": bp(kmv), "TE": bp(kone["TE"]), "EB": bp(kone["EB"])}
    # weight x normalisation planes of the MV sum and the plain normalisations (half-plane host arrays)
    w = q.mv_weights(ests)
    L = q.modl_h
    fn_mv = {XY: _full_from_half(-(L * (L + 1.) / 2.) * q.AL[XY] * q.mask_K * w[XY], N) for XY in ests}
    fn_one = {XY: _full_from_half(-(L * (L + 1.) / 2.) * q.AL[XY] * q.mask_K, N) for XY in ("TE", "EB")}
    kfull = {X: e.hc_to_full(k64[i]).cpu().numpy() for i, X in enumerate("TEB")}
    del kmv, kone, k64, q, c
    _POL_CACHE.clear()
    torch.cuda.empty_cache()
    mo.set_workers(min(64, os.cpu_count() or 16))
    try:
        cl = {k: th.lCl(k, ml) for k in ("TT", "EE", "BB", "TE")}
        qr = qo.QEOracle(shape, g.step_y, g.step_x, cl, dict(T=nT, P=nP), beam, dict(T=tmask, P=tmask), kmask_K=kmask)
        bo = so.bin2D(ml, edges)
        fo = mo.FourierCalc(shape, g.step_y, g.step_x)
        acc = np.zeros(shape, dtype=np.complex128)
        ref = {}
        for XY in ests:
            u = qr.unnormalized_ft(XY, kfull[XY[0]], kfull[XY[1]])
            acc += fn_mv[XY] * u
            if XY in fn_one:
                kr = fn_one[XY] * u
                ref[XY] = bo.bin(fo.f2power(kr, kr))[1]
            del u
        ref["MV"] = bo.bin(fo.f2power(acc, acc))[1]
    finally:
        mo.set_workers(1)
    for key in ("MV", "TE", "EB"):
        err = np.max(np.abs(got[key] / ref[key] - 1))
        assert err < 1e-9, "%s at 8192^2 f64: bandpowers differ from the NumPy oracle arithmetic by %.3g" % (key, err)
