"""One-call C-ABI entries (include/orphics_amd.h) driven the way a non-torch host would:
NumPy arrays, device memory from oa_malloc / oa_memcpy, raw pointers -- compared with the Python classes (which are
themselves checked against the oracle elsewhere)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


class Dev(object):
    """device buffers through the library's own allocator (no torch involved)"""

    def __init__(self, lib, check):
        self.lib, self.check, self.ptrs = lib, check, []

    def up(self, a):
        a = np.ascontiguousarray(a)
        p = ctypes.c_void_p()
        self.check(self.lib.oa_malloc(ctypes.byref(p), a.nbytes))
        self.check(self.lib.oa_memcpy(p, a.ctypes.data_as(ctypes.c_void_p), a.nbytes, 1, None))
        self.ptrs.append(p)
        return p

    def zeros(self, nbytes):
        p = ctypes.c_void_p()
        self.check(self.lib.oa_malloc(ctypes.byref(p), nbytes))
        self.check(self.lib.oa_memset(p, 0, nbytes, None))
        self.ptrs.append(p)
        return p

    def down(self, p, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        self.check(self.lib.oa_memcpy(out.ctypes.data_as(ctypes.c_void_p), p, out.nbytes, 2, None))
        self.check(self.lib.oa_stream_synchronize(None))
        return out

    def free(self):
        for p in self.ptrs:
            self.lib.oa_free(p)


def _setup(N=256, res=2.0):
    from orphics_amd import cosmology, lensing, maps
    from orphics_amd.geometry import FlatGeometry
    shape = (N, N)
    g = FlatGeometry.from_res(shape, res)
    th = cosmology.default_theory()
    ml = g.modlmap()
    beam = maps.gauss_beam(ml, 1.5)
    noise = np.full(shape, cosmology.white_noise_power(1.0))
    tmask = maps.mask_kspace(shape, g, lmin=300, lmax=2000)
    kmask = maps.mask_kspace(shape, g, lmin=20, lmax=3500)
    q = lensing.qest(shape, g, th, noise2d=noise, beam2d=beam, kmask=tmask, kmask_K=kmask, unlensed_equals_lensed=True, dtype="f32")
    rng = np.random.default_rng(4)
    cl = th.lCl("TT", ml)
    tk = np.fft.fft2(rng.standard_normal(shape)) * np.sqrt((cl * beam ** 2 + noise) / g.pixarea)
    tmap = np.fft.ifft2(tk).real.astype(np.float32)
    return q, g, tmap, (cl * beam ** 2 + noise)


def test_one_call_entries_from_raw_pointers():
    from orphics_amd import _lib, maps, mc
    from orphics_amd._lib import check
    lib = _lib.load()
    N = 256
    q, g, tmap, tot = _setup(N)
    kp = q.eng.kp
    dev = Dev(lib, check)
    try:
        plan = ctypes.c_void_p()
        check(lib.oa_plan_create(N, N, _lib.OA_F32, ctypes.byref(plan)))
        ly, lx = g.laxes()
        check(lib.oa_plan_set_laxes(plan, ly.ctypes.data_as(ctypes.c_void_p), lx.ctypes.data_as(ctypes.c_void_p)))
        FG, FH, Fn = [dev.up(t.cpu().numpy()) for t in q._F["TT"]]
        check(lib.oa_plan_set_filters(plan, FG, FH, Fn, q.leg_cols, q.kappa_cols, q.leg_rows, q.kappa_rows, -1))
        d_map = dev.up(tmap)
        d_out = dev.up(np.full((N, kp), 7 + 7j, dtype=np.complex64))        # garbage: zero_outside must clear it
        # --- oa_qe_tt from a real map == Estimator.reconstruct_tt_from_map
        check(lib.oa_qe_tt(plan, d_map, None, None, d_out, 1, None))
        got = dev.down(d_out, (N, kp), np.complex64)
        want = q.reconstruct_tt_from_map(q.eng.to_real(tmap)).cpu().numpy()
        assert np.array_equal(got, want)                                   # same kernels, same launch geometry
        assert np.all(got[:, q.kappa_cols:] == 0)
        # --- Fourier-space legs (split estimator contract: X and Y legs differ)
        k1 = q.eng.rfft(q.eng.to_real(tmap))
        k2 = q.eng.rfft(q.eng.to_real(tmap[::-1].copy()))
        dk1, dk2 = dev.up(k1.cpu().numpy()), dev.up(k2.cpu().numpy())
        check(lib.oa_qe_tt(plan, None, dk1, dk2, d_out, 1, None))
        got = dev.down(d_out, (N, kp), np.complex64)
        assert np.array_equal(got, q.reconstruct_tt_hc(k1, k2).cpu().numpy())
        # --- plan-owned output plane
        check(lib.oa_qe_tt(plan, d_map, None, None, None, 0, None))
        own = dev.down(ctypes.c_void_p(lib.oa_plan_kappa(plan)), (N, kp), np.complex64)
        assert np.array_equal(own, want)
        # --- bins + one Monte-Carlo step per call
        edges = np.linspace(100, 3000, 12)
        nids = edges.size + 1
        ids = q.eng.modl_digitize(torch.as_tensor(edges, device=q.eng.device), half=True)
        d_ids = dev.up(ids.cpu().numpy())
        norm = g.area / float(N * N) ** 2
        check(lib.oa_plan_set_bins(plan, d_ids, nids, norm, None))
        dd = nids - 2
        n, S, C = dev.zeros(8), dev.zeros(8 * dd), dev.zeros(8 * dd * dd)
        for _ in range(3):
            check(lib.oa_qe_tt_moments(plan, d_map, n, S, C, None))
        hn, hS, hC = dev.down(n, (1,), np.int64), dev.down(S, (dd,), np.float64), dev.down(C, (dd, dd), np.float64)
        counts = dev.down(ctypes.c_void_p(lib.oa_plan_bin_counts(plan)), (nids,), np.int64)
        sums, cfull = q.eng.bin_power(torch.as_tensor(want, device=q.eng.device), torch.as_tensor(want, device=q.eng.device), norm, ids, nids, herm=True)
        assert np.array_equal(counts, cfull.cpu().numpy())
        b = (sums[1:-1] / cfull[1:-1]).cpu().numpy()
        assert hn[0] == 3
        np.testing.assert_allclose(hS, 3 * b, rtol=1e-12)
        np.testing.assert_allclose(hC, 3 * np.outer(b, b), rtol=1e-12)
        # --- oa_mc_run == the Python driver (same Philox streams, deterministic binning)
        tot_h = tot[:, :N // 2 + 1]
        drv = mc.GaussianN0MonteCarlo(q, tot_h, edges, comm=None, base_seed=11, mean_field=True)
        st = drv.run(9)
        check(lib.oa_memset(n, 0, 8, None)); check(lib.oa_memset(S, 0, 8 * dd, None)); check(lib.oa_memset(C, 0, 8 * dd * dd, None))
        mf = dev.zeros(8 * 2 * N * kp)
        d_cs = dev.up(drv.cs.cpu().numpy())
        check(lib.oa_mc_run(plan, 11, 0, 4, d_cs, n, S, C, mf, None))
        check(lib.oa_mc_run(plan, 11, 4, 9, d_cs, n, S, C, mf, None))
        assert dev.down(n, (1,), np.int64)[0] == 9 == st.count("n0")
        np.testing.assert_allclose(dev.down(S, (dd,), np.float64) / 9, st.mean("n0"), rtol=1e-13)
        np.testing.assert_allclose(dev.down(mf, (N, kp, 2), np.float64), st.stack_sum("mf"), rtol=0, atol=0)
        # --- oa_filter_map == maps.filter_map
        filt = maps.gauss_beam(g.modlmap(), 3.0)
        fh = np.zeros((N, kp), dtype=np.float32)
        fh[:, :N // 2 + 1] = filt[:, :N // 2 + 1]
        d_f, d_o = dev.up(fh), dev.zeros(4 * N * N)
        check(lib.oa_filter_map(plan, d_map, d_f, d_o, None))
        np.testing.assert_allclose(dev.down(d_o, (N, N), np.float32), maps.filter_map(tmap, filt), rtol=0, atol=2e-5 * np.abs(tmap).max())
        # --- single-rank RCCL communicator: all-reduce is the identity
        uid = (ctypes.c_char * 128)()
        check(lib.oa_comm_unique_id(uid))
        comm = ctypes.c_void_p()
        check(lib.oa_comm_init(1, 0, uid, ctypes.byref(comm)))
        check(lib.oa_allreduce(comm, S, dd, 0, None))
        check(lib.oa_allreduce(comm, n, 1, 1, None))
        check(lib.oa_stream_synchronize(None))
        assert dev.down(n, (1,), np.int64)[0] == 9
        check(lib.oa_comm_destroy(comm))
        # --- misuse is refused with a message, not a crash
        assert lib.oa_qe_tt(plan, d_map, dk1, None, d_out, 1, None) != 0 and b"either" in lib.oa_last_error()
        check(lib.oa_plan_destroy(plan))
    finally:
        dev.free()


def test_one_call_pol_estimator_matches_fine_grained():
    """One polarised estimator in one call -- oa_qe_mv with one estimator (Estimator.reconstruct_hc) and oa_qe_pol (called
    directly) -- == the chain of fine-grained calls they replace."""
    from orphics_amd import cosmology, lensing, maps
    from orphics_amd.geometry import FlatGeometry
    N = 128
    shape = (N, N)
    g = FlatGeometry.from_res(shape, 2.0)
    th = cosmology.default_theory()
    ml = g.modlmap()
    beam = maps.gauss_beam(ml, 1.5)
    noise = np.full(shape, cosmology.white_noise_power(1.0))
    tm = maps.mask_kspace(shape, g, lmin=300, lmax=2500)
    q = lensing.qest(shape, g, th, noise2d=noise, beam2d=beam, kmask=tm, noise2d_P=2 * noise, kmask_P=tm, kmask_K=None,
                     pol=True, unlensed_equals_lensed=True, dtype="f64")
    e = q.eng
    rng = np.random.default_rng(1)
    kE = e.rfft(e.to_real(rng.standard_normal(shape)))
    kB = e.rfft(e.to_real(rng.standard_normal(shape)))
    for XY in ("EB", "TE", "EE"):
        got = q.reconstruct_hc(XY, kE, kB).cpu().numpy()
        G = q._setup_general(XY)
        ax, ay = e.hc(), e.hc()
        s0 = 1.0 / float(e.npix) ** 2
        for i, (sign, FG, FH, swap) in enumerate(G["pieces"]):
            kg, kh = (kB, kE) if swap else (kE, kB)
            c = e.qe_legs_cols(kg, kh, FG, FH, out=(e.hc(), e.hc(), e.hc()), width=G["wl"], rband=G["rl"])
            e.qe_rows(c[0], c[1], c[2], ax, ay, scale=sign * s0, accumulate=(i > 0), win=G["wl"], wout=G["wk"], mrow=q.mrow)
        want = e.qe_cols_div(ax, ay, G["Fnorm"], width=G["wk"], rband=G["rk"]).cpu().numpy()
        w = N // 2 + 1
        assert np.abs(got[:, :w] - want[:, :w]).max() <= 1e-14 * np.abs(want).max()
        from orphics_amd._lib import check
        from orphics_amd.engine import _ptr, _stream
        n, signs, fgs, fhs, swaps = G["c_args"]
        pol = e.hc()
        check(e.lib.oa_qe_pol(e.plan, n, signs, fgs, fhs, swaps, _ptr(kE), _ptr(kB), _ptr(G["Fnorm"]), _ptr(pol), 0, int(G["wl"]), int(G["wk"]),
                              int(G["rl"]), int(G["rk"]), int(q.mrow), 0, _stream()))
        assert np.array_equal(pol.cpu().numpy()[:, :w], got[:, :w])           # same kernels on the same operands


@pytest.mark.parametrize("N,res,prec", [(4096, 0.5, "f32"), (4096, 0.5, "f64"), (1024, 1.0, "f32")])
def test_two_maps_per_call_equals_two_calls(N, res, prec):
    """oa_qe_tt_moments2: two realisations sharing every launch behind their row transforms (4096-point columns: the
    fused pair path; 1024: the sequential fall-back) accumulate exactly what two oa_qe_tt_moments calls do."""
    from orphics_amd import cosmology, lensing, maps
    from orphics_amd.geometry import FlatGeometry
    shape = (N, N)
    g = FlatGeometry.from_res(shape, res)
    th = cosmology.default_theory()
    ml = g.modlmap()
    q = lensing.qest(shape, g, th, noise2d=np.full(shape, cosmology.white_noise_power(1.0)), beam2d=maps.gauss_beam(ml, 1.5),
                     kmask=maps.mask_kspace(shape, g, lmin=300, lmax=2000), kmask_K=maps.mask_kspace(shape, g, lmin=20, lmax=3500),
                     unlensed_equals_lensed=True, dtype=prec)
    e = q.eng
    edges = torch.as_tensor(np.linspace(20, 3500, 20), device=e.device)
    ids = e.modl_digitize(edges, half=True)
    q.bind_bins(ids, 21, g.area / float(N * N) ** 2)
    m0 = e.irfft(e.grf_hc(5, 0), scale=1.0 / N)
    m1 = e.irfft(e.grf_hc(5, 1), scale=1.0 / N)

    def acc():
        return (torch.zeros(1, dtype=torch.int64, device=e.device), torch.zeros(19, dtype=torch.float64, device=e.device),
                torch.zeros(19, 19, dtype=torch.float64, device=e.device))
    a, b = acc(), acc()
    q.tt_moments(m0, *a); q.tt_moments(m1, *a)
    q.tt_moments2(m0, m1, *b)
    q.tt_moments2(m1, m0, *b); q.tt_moments(m0, *a); q.tt_moments(m1, *a)      # again: work planes are reused correctly
    torch.cuda.synchronize()
    assert int(a[0]) == int(b[0]) == 4
    assert float(a[1].abs().min()) > 0
    np.testing.assert_allclose(b[1].cpu().numpy(), a[1].cpu().numpy(), rtol=1e-13)
    np.testing.assert_allclose(b[2].cpu().numpy(), a[2].cpu().numpy(), rtol=1e-12)


@pytest.mark.parametrize("N,res,prec", [(4096, 0.5, "f32"), (2048, 1.0, "f64"), (512, 2.0, "f32")])
def test_mc_run_in_batches_equals_one_by_one(N, res, prec):
    """oa_mc_run hands OA_OPT_MC_BATCH realisations to every launch (grid z: GRF draw, leg planes, inverse pass 2, row stage,
    divergence); the moments and the mean-field stack are those of the one-realisation-per-launch loop (same kernels on the
    same operands, accumulated in the same order).  512^2: geometry without the batched row stage -> falls back."""
    import os
    from orphics_amd import cosmology, lensing, maps, mc
    from orphics_amd.geometry import FlatGeometry
    shape = (N, N)
    g = FlatGeometry.from_res(shape, res)
    th = cosmology.default_theory()
    ml = g.modlmap()
    beam = maps.gauss_beam(ml, 1.5)
    noise = np.full(shape, cosmology.white_noise_power(1.0))
    q = lensing.qest(shape, g, th, noise2d=noise, beam2d=beam, kmask=maps.mask_kspace(shape, g, lmin=300, lmax=2000),
                     kmask_K=maps.mask_kspace(shape, g, lmin=20, lmax=3500), unlensed_equals_lensed=True, dtype=prec)
    tot = (th.lCl("TT", ml) * beam ** 2 + noise)[:, :N // 2 + 1]
    edges = np.linspace(20, 3500, 20)
    res_ = {}
    for batch in ("1", "4", "3", "6"):
        q.eng.set_option("mc_batch", int(batch))
        try:
            drv = mc.GaussianN0MonteCarlo(q, tot, edges, comm=None, base_seed=21, mean_field=True)
            st = drv.run(11)                                       # 11 = 4 + 4 + 3 (batch 4), 6 + 5, 3 + 3 + 3 + 2, ...
            res_[batch] = (st.count("n0"), np.array(st.mean("n0")), np.array(st.cov("n0")), st.stack_sum("mf").copy())
        finally:
            q.eng.set_option("mc_batch", 0)
    n1, m1, c1, s1 = res_["1"]
    assert n1 == 11 and np.all(m1 > 0)
    for batch in ("4", "3", "6"):
        nb, mb, cb, sb = res_[batch]
        assert nb == 11
        np.testing.assert_allclose(mb, m1, rtol=1e-13)
        np.testing.assert_allclose(cb, c1, rtol=1e-9, atol=1e-13 * np.abs(c1).max())
        assert np.array_equal(sb, s1)


def test_round3_entries_from_raw_pointers():
    """The entries added in round 3, driven with raw pointers only: oa_fft_c2r_windowed, oa_hc_derivs + oa_lens_taylor (all
    Taylor terms of the lensing op in two launches), oa_mc_run_windowed (window == 1 reproduces oa_mc_run), oa_probe_copy /
    oa_probe_read, oa_plan_rsplit -- against NumPy on the host."""
    from orphics_amd import _lib
    from orphics_amd.engine import _ptr
    lib = _lib.load()
    check = _lib.check
    q, g, tmap, tot = _setup(256, 2.0)
    N = 256
    e = q._bind()                                   # plan with laxes and the TT filters
    d = Dev(lib, check)
    try:
        kp = e.kp
        rng = np.random.default_rng(5)
        # windowed C2R
        x = rng.standard_normal((N, N)).astype(np.float32)
        w = rng.uniform(0, 1, (N, N)).astype(np.float32)
        khc = np.zeros((N, kp), dtype=np.complex64)
        khc[:, :N // 2 + 1] = np.fft.rfft2(x)
        out = d.zeros(N * N * 4)
        check(lib.oa_fft_c2r_windowed(e.plan, d.up(khc), out, 1.0 / (N * N), d.up(w), None))
        got = d.down(out, (N, N), np.float32)
        assert np.abs(got - x * w).max() < 2e-5
        # derivative planes + one-pass Taylor gather (order 3: planes (1,0), (0,1), (2,0), (1,1), (0,2))
        ly, lx = g.laxes()
        lyd, lxd = ly.copy(), lx.copy()
        lyd[N // 2] = 0
        lxd[N // 2] = 0
        nd = 5
        dk = d.zeros(nd * N * kp * 8)
        check(lib.oa_hc_derivs(e.plan, d.up(khc), 3, dk, N * kp, None))
        planes = d.down(dk, (nd, N, kp), np.complex64)[:, :, :N // 2 + 1]
        k0 = np.fft.rfft2(x.astype(np.float64))
        ilx, ily = 1j * lxd[None, :N // 2 + 1], 1j * lyd[:, None]
        want = [k0 * ilx, k0 * ily, k0 * ilx ** 2, k0 * ilx * ily, k0 * ily ** 2]          # idx(a, b) = n (n + 1) / 2 - 1 + b
        for i in range(nd):
            assert np.abs(planes[i] - want[i]).max() < 3e-6 * np.abs(want[i]).max()
        dr = np.stack([np.fft.irfft2(wk, s=(N, N)) for wk in want]).astype(np.float32)
        sx = rng.integers(-3, 4, (N, N)).astype(np.int32)
        sy = rng.integers(-3, 4, (N, N)).astype(np.int32)
        ddx = (rng.uniform(-0.5, 0.5, (N, N)) * abs(g.step_x)).astype(np.float32)
        ddy = (rng.uniform(-0.5, 0.5, (N, N)) * abs(g.step_y)).astype(np.float32)
        lens = d.zeros(N * N * 4)
        check(lib.oa_lens_taylor(e.plan, d.up(x), d.up(dr), N * N, 3, d.up(sx), d.up(sy), d.up(ddx), d.up(ddy), lens, None))
        got = d.down(lens, (N, N), np.float32)
        yy, xx = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
        gy, gx = (yy + sy) % N, (xx + sx) % N
        ref = (x[gy, gx] + ddx * dr[0][gy, gx] + ddy * dr[1][gy, gx] + 0.5 * ddx ** 2 * dr[2][gy, gx] + ddx * ddy * dr[3][gy, gx]
               + 0.5 * ddy ** 2 * dr[4][gy, gx])
        assert np.abs(got - ref).max() < 1e-5 * np.abs(ref).max()
        # windowed Monte-Carlo shard with window == 1 against the plain shard (same Philox draws on the leg band)
        edges = np.linspace(100, 3000, 8)
        ids = e.modl_digitize(torch.as_tensor(edges, device=e.device), half=True)
        q.bind_bins(ids, len(edges) + 1, g.area / float(N * N) ** 2)
        e = q._bind_bins()
        dd = len(edges) - 1
        amp = np.zeros((N, kp), dtype=np.float32)
        amp[:, :N // 2 + 1] = np.sqrt(tot[:, :N // 2 + 1] * float(N * N) ** 2 / g.area)
        cs = d.up(amp)
        res = []
        for windowed in (False, True):
            n, S, C = d.zeros(8), d.zeros(8 * dd), d.zeros(8 * dd * dd)
            if windowed:
                check(lib.oa_mc_run_windowed(e.plan, 7, 0, 5, cs, d.up(np.ones((N, N), dtype=np.float32)), n, S, C, None, None))
            else:
                check(lib.oa_mc_run(e.plan, 7, 0, 5, cs, n, S, C, None, None))
            assert int(d.down(n, (1,), np.int64)[0]) == 5
            res.append(d.down(S, (dd,), np.float64))
        assert np.abs(res[1] / res[0] - 1).max() < 2e-4
        # bandwidth probes run and preserve the data; the R-split query answers for this (small) geometry
        nb = 1 << 20
        a = d.up(np.arange(nb // 4, dtype=np.int32))
        b = d.zeros(nb)
        check(lib.oa_probe_copy(b, a, nb, None))
        assert np.array_equal(d.down(b, (nb // 4,), np.int32), np.arange(nb // 4, dtype=np.int32))
        check(lib.oa_probe_read(a, nb, d.zeros(8 << 20), None))
        assert lib.oa_plan_rsplit(e.plan) in (0, 4)
        with pytest.raises(_lib.OrphicsAmdError):
            check(lib.oa_hc_derivs(e.plan, a, 9, dk, N * kp, None))               # order out of range: loud
    finally:
        d.free()


@pytest.mark.parametrize("N,res,prec", [(4096, 0.5, "f32"), (4096, 0.5, "f64"), (2048, 1.0, "f32")])
def test_binning_in_the_divergence_launch_equals_the_separate_histogram(N, res, prec):
    """oa_qe_tt_moments / _moments2 with the radial histogram and the moment update in the tail of the single-pass divergence
    kernel (fft_divbin.hpp) against the same calls with plan option div_bin = 0 (bin_kernel + bin_final_kernel over the kappa plane):
    same per-mode arithmetic, other order of the float64 sums -> 1e-13; and against bin2D-style bandpowers of the kappa plane."""
    import os
    from orphics_amd import cosmology, lensing, maps
    from orphics_amd.geometry import FlatGeometry
    shape = (N, N)
    g = FlatGeometry.from_res(shape, res)
    th = cosmology.default_theory()
    ml = g.modlmap()
    q = lensing.qest(shape, g, th, noise2d=np.full(shape, cosmology.white_noise_power(1.0)), beam2d=maps.gauss_beam(ml, 1.5),
                     kmask=maps.mask_kspace(shape, g, lmin=300, lmax=2000), kmask_K=maps.mask_kspace(shape, g, lmin=20, lmax=3500),
                     unlensed_equals_lensed=True, dtype=prec)
    e = q.eng
    edges = torch.as_tensor(np.linspace(20, 3500, 20), device=e.device)
    ids = e.modl_digitize(edges, half=True)
    q.bind_bins(ids, 21, g.area / float(N * N) ** 2)
    eb = q._bind_bins()
    m = [e.irfft(e.grf_hc(9, i), scale=1.0 / N) for i in range(3)]

    def acc():
        return (torch.zeros(1, dtype=torch.int64, device=e.device), torch.zeros(19, dtype=torch.float64, device=e.device),
                torch.zeros(19, 19, dtype=torch.float64, device=e.device))
    assert eb.lib.oa_plan_div_fused(eb.plan) == 1
    fused = acc()
    q.tt_moments(m[0], *fused); q.tt_moments2(m[1], m[2], *fused); q.tt_moments(m[1], *fused)
    torch.cuda.synchronize()
    eb.set_option("div_bin", 0)
    try:
        assert eb.lib.oa_plan_div_fused(eb.plan) == 0
        sep = acc()
        q.tt_moments(m[0], *sep); q.tt_moments2(m[1], m[2], *sep); q.tt_moments(m[1], *sep)
        torch.cuda.synchronize()
    finally:
        eb.set_option("div_bin", 1)
    assert int(fused[0]) == int(sep[0]) == 4
    assert float(sep[1].abs().min()) > 0
    np.testing.assert_allclose(fused[1].cpu().numpy(), sep[1].cpu().numpy(), rtol=1e-13)
    np.testing.assert_allclose(fused[2].cpu().numpy(), sep[2].cpu().numpy(), rtol=1e-12)
    # one map against the public fine-grained calls: kappa plane -> binned auto power
    one = acc()
    q.tt_moments(m[2], *one)
    kap = q.reconstruct_tt_hc(e.rfft(m[2]))
    sums, counts = e.bin_power(kap, kap, g.area / float(N * N) ** 2, ids, 21, herm=True)
    want = (sums / counts.to(torch.float64))[1:-1].cpu().numpy()
    np.testing.assert_allclose(one[1].cpu().numpy(), want, rtol=(2e-6 if prec == "f32" else 1e-11))


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_ticket_tail_stress_on_three_streams_is_bit_reproducible(prec):
    """The last-workgroup hand-over of the fused binning tail (fft_divbin.hpp: write-through partial sums, ticket counter, no
    device fence) under load: 3 x 3400 = 10 200 one-call moment launches on three HIP streams (three forked handles: own
    plans, own tickets, shared read-only filters) without any host synchronisation, alternating one- and two-map calls over
    a pool of maps.  Every lane's (n, S, C) must equal BIT FOR BIT the same launch sequence issued serially with a device
    synchronisation after each call: a stale partial sum read by a last workgroup would change low bits of S.  The serial
    fused moments are tied to the separate-histogram path (plan option div_bin = 0: other summation order) at 1e-13."""
    from orphics_amd import cosmology, lensing, maps
    from orphics_amd.geometry import FlatGeometry
    N, res, per_lane = 2048, 1.0, 3400
    shape = (N, N)
    g = FlatGeometry.from_res(shape, res)
    th = cosmology.default_theory()
    ml = g.modlmap()
    q0 = lensing.qest(shape, g, th, noise2d=np.full(shape, cosmology.white_noise_power(1.0)), beam2d=maps.gauss_beam(ml, 1.5),
                      kmask=maps.mask_kspace(shape, g, lmin=300, lmax=2000), kmask_K=maps.mask_kspace(shape, g, lmin=20, lmax=3000),
                      unlensed_equals_lensed=True, dtype=prec)
    e0 = q0.eng
    edges = torch.as_tensor(np.linspace(40, 2900, 20), device=e0.device)
    ids = e0.modl_digitize(edges, half=True)
    lanes = [q0, q0.fork(), q0.fork()]
    for q in lanes:
        q.bind_bins(ids, 21, g.area / float(N * N) ** 2)
        assert q._bind_bins().lib.oa_plan_div_fused(q.eng.plan) == 1
    pool = [e0.irfft(e0.grf_hc(31, i), scale=1.0 / N) for i in range(5)]

    def acc():
        return (torch.zeros(1, dtype=torch.int64, device=e0.device), torch.zeros(19, dtype=torch.float64, device=e0.device),
                torch.zeros(19, 19, dtype=torch.float64, device=e0.device))

    def issue(q, a, k):
        # call k of a lane: two-map call every third time, maps walking the pool at lane-independent strides
        if k % 3 == 2:
            q.tt_moments2(pool[k % 5], pool[(3 * k + 1) % 5], *a)
        else:
            q.tt_moments(pool[(2 * k) % 5], *a)
    # reference: the same sequences one lane after the other, with a device synchronisation after EVERY call (S and C are
    # float sums in launch order, so the reference runs the full length too)
    ref = []
    for q in lanes:
        a = acc()
        for k in range(per_lane):
            issue(q, a, k)
            torch.cuda.synchronize()
        ref.append([t.clone() for t in a])
    # stress: three streams, round-robin issue, no synchronisation until the end
    streams = [torch.cuda.Stream() for _ in lanes]
    accs = [acc() for _ in lanes]
    torch.cuda.synchronize()
    for k in range(per_lane):
        for q, st, a in zip(lanes, streams, accs):
            with torch.cuda.stream(st):
                issue(q, a, k)
    torch.cuda.synchronize()
    for j, (a, r) in enumerate(zip(accs, ref)):
        assert int(a[0]) == int(r[0]) == per_lane + per_lane // 3
        assert torch.equal(a[1], r[1]), "lane %d: S differs by %g" % (j, float((a[1] - r[1]).abs().max() / r[1].abs().max()))
        assert torch.equal(a[2], r[2]), "lane %d: C differs" % j
    # the serial fused moments against the separate histogram launches (another order of the float64 sums)
    lanes[0].eng.set_option("div_bin", 0)
    try:
        sep = acc()
        for k in range(60):
            issue(lanes[0], sep, k)
        torch.cuda.synchronize()
    finally:
        lanes[0].eng.set_option("div_bin", 1)
    fus = acc()
    for k in range(60):
        issue(lanes[0], fus, k)
    torch.cuda.synchronize()
    np.testing.assert_allclose(fus[1].cpu().numpy(), sep[1].cpu().numpy(), rtol=1e-13)
    np.testing.assert_allclose(fus[2].cpu().numpy(), sep[2].cpu().numpy(), rtol=1e-12)


@pytest.mark.parametrize("N,res,prec", [(4096, 0.5, "f32"), (4096, 0.5, "f64"), (512, 2.0, "f64"), (2048, 1.0, "f32")])
def test_windowed_mc_fused_row_pass_equals_the_two_pass_flow(N, res, prec):
    """oa_mc_run_windowed with the fused row pass (inverse columns, then C2R x window -> R2C per row in ONE kernel, real map in LDS
    only; plan option win_fused = 1, the default) against the flow it replaces (C2R with the window at its store -> real map in
    HBM -> row R2C of the from-map estimator path): same draws, same window, same estimator arithmetic up to the order of the row
    transforms' butterflies -> bandpower moments and mean-field stack agree to rounding.  (The two-pass flow is itself compared
    with the NumPy oracle on the same maps in test_windowed_monte_carlo_mean_field_matches_oracle_on_the_same_maps.)"""
    from orphics_amd import cosmology, lensing, maps, mc
    from orphics_amd.geometry import FlatGeometry
    shape = (N, N)
    g = FlatGeometry.from_res(shape, res)
    th = cosmology.default_theory()
    ml = g.modlmap()
    beam = maps.gauss_beam(ml, 1.5)
    noise = np.full(shape, cosmology.white_noise_power(1.0))
    q = lensing.qest(shape, g, th, noise2d=noise, beam2d=beam, kmask=maps.mask_kspace(shape, g, lmin=300, lmax=2000),
                     kmask_K=maps.mask_kspace(shape, g, lmin=20, lmax=3500), unlensed_equals_lensed=True, dtype=prec)
    tot = (th.lCl("TT", ml) * beam ** 2 + noise)[:, :N // 2 + 1]
    edges = np.linspace(20, 3500, 20)
    taper, _ = maps.get_taper(shape, g, taper_percent=12.0, pad_percent=3.0)
    res_ = {}
    for fused in (1, 0):
        q.eng.set_option("win_fused", fused)
        try:
            st = mc.GaussianN0MonteCarlo(q, tot, edges, comm=None, base_seed=77, mean_field=True, window=taper).run(7)
            res_[fused] = (st.count("n0"), np.array(st.mean("n0")), np.array(st.cov("n0")), st.stack_sum("mf").copy())
        finally:
            q.eng.set_option("win_fused", 1)
    (n1, m1, c1, s1), (n0, m0, c0, s0) = res_[1], res_[0]
    assert n1 == n0 == 7 and np.all(m0 > 0)
    tol = 1e-10 if prec == "f64" else 3e-5
    np.testing.assert_allclose(m1, m0, rtol=tol)
    assert np.abs(s1 - s0).max() < tol * np.abs(s0).max()

def test_rebinding_filters_at_the_same_addresses_repacks_the_divergence_tables():
    """the tile-major copies of Fnorm and of the bin ids that the fused divergence + binning launch reads were cached
    by the planes' ADDRESSES.  A second estimator of the same shape built after the first was freed gets the same addresses from the
    caching allocator (here forced: the new filters are written INTO the old planes and bound again), and the fused path then used the
    previous estimator's normalisation.  The cache is keyed on a bind generation now: every oa_plan_set_filters / oa_plan_set_bins
    repacks.  Fused (div_bin = 1) against the separate histogram (div_bin = 0) after the re-bind."""
    from orphics_amd import cosmology, lensing, maps
    from orphics_amd.geometry import FlatGeometry
    N, res = 4096, 0.5
    shape = (N, N)
    g = FlatGeometry.from_res(shape, res)
    th = cosmology.default_theory()
    ml = g.modlmap()

    def make(noise_uk):
        return lensing.qest(shape, g, th, noise2d=np.full(shape, cosmology.white_noise_power(noise_uk)), beam2d=maps.gauss_beam(ml, 1.5),
                            kmask=maps.mask_kspace(shape, g, lmin=300, lmax=2000), kmask_K=maps.mask_kspace(shape, g, lmin=20, lmax=3500),
                            unlensed_equals_lensed=True, dtype="f32")
    q = make(1.0)
    e = q.eng
    edges = torch.as_tensor(np.linspace(20, 3500, 20), device=e.device)
    ids = e.modl_digitize(edges, half=True)
    q.bind_bins(ids, 21, g.area / float(N * N) ** 2)
    eb = q._bind_bins()
    assert eb.lib.oa_plan_div_fused(eb.plan) == 1
    m = e.irfft(e.grf_hc(9, 0), scale=1.0 / N)

    def run():
        acc = (torch.zeros(1, dtype=torch.int64, device=e.device), torch.zeros(19, dtype=torch.float64, device=e.device),
               torch.zeros(19, 19, dtype=torch.float64, device=e.device))
        q.tt_moments(m, *acc)
        torch.cuda.synchronize()
        return acc[1].cpu().numpy().copy()
    first = run()
    # a different estimator's filters, written into the SAME device planes and bound again (what a caching allocator produces when the
    # first estimator is freed and a second one of the same shape is built)
    q2 = make(6.0)
    for a, b in zip(q._F["TT"], q2._F["TT"]):
        a.copy_(b)
    e._pipe_owner = None                                     # the host mirror re-binds (same pointers, new contents)
    second = run()
    assert np.abs(second / first - 1).max() > 1e-2          # the new noise level changes the normalisation
    # ... and what comes out is the second estimator's result: its OWN planes through the same plan (this also covers the packed
    # (FG, FH) table of the R-split column stage, which both the fused and the separate-histogram runs below would share if stale)
    acc2 = (torch.zeros(1, dtype=torch.int64, device=e.device), torch.zeros(19, dtype=torch.float64, device=e.device),
            torch.zeros(19, 19, dtype=torch.float64, device=e.device))
    q2.bind_bins(ids, 21, g.area / float(N * N) ** 2)
    q2.tt_moments(m, *acc2)
    torch.cuda.synchronize()
    np.testing.assert_allclose(second, acc2[1].cpu().numpy(), rtol=1e-12)
    e._pipe_owner = None
    eb.set_option("div_bin", 0)
    try:
        sep = run()
    finally:
        eb.set_option("div_bin", 1)
    np.testing.assert_allclose(second, sep, rtol=1e-12)
