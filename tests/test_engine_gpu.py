"""GPU parity of the raw C-ABI kernels (through orphics_amd.engine) against
NumPy on the same seeded inputs.  Tolerances: float64 plans 1e-12 relative to
the plane RMS; float32 plans 2e-6 (FFT) -- bin ids and counts bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def eng(ny, nx, prec):
    from orphics_amd.engine import Engine
    return Engine.get(ny, nx, prec)


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


TOL = {"f32": 3e-6, "f64": 1e-12}
SIZES = [(32, 32), (64, 64), (128, 256), (256, 64), (1024, 1024), (2048, 512)]


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("ny,nx", SIZES)
def test_rfft_irfft(ny, nx, prec):
    e = eng(ny, nx, prec)
    rng = np.random.default_rng(ny * 3 + nx)
    x = rng.standard_normal((ny, nx))
    xd = e.to_real(x)
    k = e.rfft(xd)
    ref = np.fft.rfft2(x)
    got = k.cpu().numpy()[:, :nx // 2 + 1]
    assert rel(got, ref) < TOL[prec]
    y = e.irfft(k).cpu().numpy()
    assert rel(y, x) < TOL[prec]
    # input preserved by c2r
    assert np.array_equal(k.cpu().numpy()[:, :nx // 2 + 1], got)
    # windowed C2R (power-of-two sides): the window rides on the last pass's store
    if e.pow2:
        w = rng.uniform(0.0, 1.0, (ny, nx))
        yw = e.irfft(k, window=e.to_real(w)).cpu().numpy()
        assert rel(yw, x * w) < TOL[prec]


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("ny,nx", SIZES[:5])
def test_cfft(ny, nx, prec):
    e = eng(ny, nx, prec)
    rng = np.random.default_rng(ny + nx)
    z = rng.standard_normal((ny, nx)) + 1j * rng.standard_normal((ny, nx))
    zd = e.to_complex(z)
    f = e.cfft(zd).cpu().numpy()
    assert rel(f, np.fft.fft2(z)) < TOL[prec]
    b = e.cfft(zd, inverse=True, scale=1.0 / (ny * nx)).cpu().numpy()
    assert rel(b, np.fft.ifft2(z)) < TOL[prec]


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_layouts_and_elementwise(prec):
    ny, nx = 64, 128
    e = eng(ny, nx, prec)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((ny, nx))
    x2 = rng.standard_normal((ny, nx))
    k = e.rfft(e.to_real(x))
    k2 = e.rfft(e.to_real(x2))
    full = e.hc_to_full(k).cpu().numpy()
    assert rel(full, np.fft.fft2(x)) < TOL[prec]
    back = e.full_to_hc(e.to_complex(np.fft.fft2(x))).cpu().numpy()[:, :nx // 2 + 1]
    assert rel(back, np.fft.rfft2(x)) < TOL[prec]
    p = e.f2power(k, k2, 0.25).cpu().numpy()[:, :nx // 2 + 1]
    ref = np.real(np.conj(np.fft.rfft2(x)) * np.fft.rfft2(x2)) * 0.25
    assert rel(p, ref) < 10 * TOL[prec]
    f = rng.uniform(0.5, 2, (ny, nx))
    fh = e.fullreal_to_hc(e.to_real(f))
    assert np.array_equal(fh.cpu().numpy()[:, :nx // 2 + 1], f[:, :nx // 2 + 1].astype(fh.cpu().numpy().dtype))
    kf = e.cmul_real(k, fh).cpu().numpy()[:, :nx // 2 + 1]
    assert rel(kf, np.fft.rfft2(x) * f[:, :nx // 2 + 1]) < TOL[prec]
    a = e.to_real(x); b = e.to_real(x2)
    assert rel(e.mul_real(a, b).cpu().numpy(), x * x2) < TOL[prec]
    assert rel(e.axpby(a, b, 2.0, -0.5).cpu().numpy(), 2 * x - 0.5 * x2) < TOL[prec]
    # even-symmetric real plane expansion
    sym = np.real(np.fft.fft2(np.fft.ifft2(f).real))  # even-symmetric plane
    sh = e.fullreal_to_hc(e.to_real(sym))
    assert rel(e.hcreal_to_full(sh).cpu().numpy(), sym) < TOL[prec]


def test_digitize_bit_exact():
    e = eng(64, 64, "f64")
    rng = np.random.default_rng(2)
    edges = np.linspace(20., 3500., 20)
    x = rng.uniform(0, 4000, 100003)
    x[:20] = edges  # exact ties go to the lower bin
    x[20] = np.nan
    ids = e.digitize(torch.as_tensor(x, device=e.device), torch.as_tensor(edges, device=e.device)).cpu().numpy()
    assert np.array_equal(ids, np.digitize(x, edges, right=True).astype(np.int32))


@pytest.mark.parametrize("ny,nx,res", [(64, 64, 2.0), (512, 512, 0.5 * 4096 / 512), (1024, 1024, 2.0)])
def test_modl_digitize_bit_exact(ny, nx, res):
    from orphics_amd.geometry import FlatGeometry
    e = eng(ny, nx, "f64")
    g = FlatGeometry.from_res((ny, nx), res)
    ly, lx = g.laxes()
    e.set_laxes(ly, lx)
    ml = g.modlmap()
    fund = abs(ly[1])
    edges = np.concatenate([[0.], fund * np.array([1, 5, 10, 13, 20, 25.]), [ml.max() * 0.9]])
    edges = np.unique(edges[edges <= ml.max() * 0.9])
    ed = torch.as_tensor(edges, device=e.device)
    ids, modl = e.modl_digitize(ed, half=False, want_modl=True)
    assert np.array_equal(modl.cpu().numpy(), ml)  # bit-exact float64 |ell|
    assert np.array_equal(ids.cpu().numpy(), np.digitize(ml.reshape(-1), edges, right=True).reshape(ny, nx))
    idh = e.modl_digitize(ed, half=True).cpu().numpy()
    assert np.array_equal(idh[:, :nx // 2 + 1], np.digitize(ml[:, :nx // 2 + 1].reshape(-1), edges, right=True).reshape(ny, nx // 2 + 1))
    assert (idh[:, nx // 2 + 1:] == -1).all()


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_bin_matches_bincount(prec):
    ny, nx = 256, 512
    e = eng(ny, nx, prec)
    rng = np.random.default_rng(9)
    nedges = 30
    nids = nedges + 1
    ids = rng.integers(0, nids, size=ny * nx).astype(np.int32)
    ids[:5000] = 3  # long runs
    data = rng.standard_normal(ny * nx)
    w = rng.uniform(0.5, 2, ny * nx)
    dd = e.to_real(data); idd = torch.as_tensor(ids, device=e.device); wd = e.to_real(w)
    dref = dd.cpu().numpy().astype(np.float64)
    wref = wd.cpu().numpy().astype(np.float64)
    s, c = e.bin(dd, idd, nids)
    assert np.array_equal(c.cpu().numpy(), np.bincount(ids, minlength=nids))
    np.testing.assert_allclose(s.cpu().numpy(), np.bincount(ids, dref, minlength=nids), rtol=1e-12, atol=1e-9)
    s, ws = e.bin(dd, idd, nids, weights=wd)
    np.testing.assert_allclose(ws.cpu().numpy(), np.bincount(ids, wref, minlength=nids), rtol=1e-12)
    np.testing.assert_allclose(s.cpu().numpy(), np.bincount(ids, dref * wref, minlength=nids), rtol=1e-10, atol=1e-9)
    # determinism: two runs bitwise equal
    s2, _ = e.bin(dd, idd, nids, weights=wd)
    assert torch.equal(s, s2)
    # squared deviations about per-id aux
    aux = rng.standard_normal(nids)
    s, _ = e.bin(dd, idd, nids, aux=torch.as_tensor(aux, device=e.device), mode=1)
    np.testing.assert_allclose(s.cpu().numpy(), np.bincount(ids, (dref - aux[ids]) ** 2, minlength=nids), rtol=1e-10)
    # NaN masking
    dn = dref.copy(); dn[::7] = np.nan
    s, c = e.bin(e.to_real(dn), idd, nids, skip_nan=True)
    keep = ~np.isnan(dn)
    assert np.array_equal(c.cpu().numpy(), np.bincount(ids[keep], minlength=nids))
    # ragged n (not a multiple of 4)
    n2 = 1003
    s, c = e.bin(dd[:n2].contiguous(), idd[:n2].contiguous(), nids)
    assert np.array_equal(c.cpu().numpy(), np.bincount(ids[:n2], minlength=nids))


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_bin_hermitian_half_plane(prec):
    """Half-plane binning with multiplicities == full-plane binning (counts exact)."""
    from orphics_amd.geometry import FlatGeometry
    ny, nx = 128, 256
    e = eng(ny, nx, prec)
    g = FlatGeometry.from_res((ny, nx), 2.0)
    ly, lx = g.laxes(); e.set_laxes(ly, lx)
    edges = np.arange(100., 3000., 200.)
    ed = torch.as_tensor(edges, device=e.device)
    rng = np.random.default_rng(4)
    x = rng.standard_normal((ny, nx))
    k = e.rfft(e.to_real(x))
    p_hc = e.f2power(k, k, 1.0)
    p_full = e.hcreal_to_full(p_hc)
    nids = len(edges) + 1
    sh, ch = e.bin(p_hc, e.modl_digitize(ed, half=True), nids, herm=True)
    sf, cf = e.bin(p_full, e.modl_digitize(ed, half=False), nids)
    assert torch.equal(ch, cf)
    np.testing.assert_allclose(sh.cpu().numpy(), sf.cpu().numpy(), rtol=1e-12)


@pytest.mark.parametrize("ny,nx,width", [(64, 8192, 300), (32, 8192, 512), (64, 8192, 1), (32, 16384, 760), (64, 16384, 37)])
def test_band_limited_r2c_row_kernels_on_short_maps(ny, nx, width):
    """The one-wave-per-row (8192-point rows) and two-waves-per-row (16384-point rows) R2C kernels on maps with few
    rows (fewer rows than resident waves) and extreme widths: rfft(width=...) equals torch.fft.rfft2 on the kept columns."""
    from orphics_amd.engine import Engine
    e = Engine.get(ny, nx, "f32")
    x = torch.randn(ny, nx, device="cuda", dtype=torch.float32)
    ref = torch.fft.rfft2(x.double())
    out = e.hc(); out[:] = 7.0
    e.rfft(x, out=out, width=width)
    err = float((out[:, :width].to(torch.complex128) - ref[:, :width]).abs().max() / ref.abs().max())
    assert err < 3e-6, err
    assert bool((out[:, width:] == 7.0).all())


def test_grf_band_is_a_subset_of_the_full_draw():
    """oa_grf_hc_band draws only the active region, with the same Philox counters as the full plane."""
    from orphics_amd.engine import Engine
    for prec in ("f32", "f64"):
        e = Engine.get(256, 512, prec)
        cs = torch.rand(256, e.kp, device="cuda", dtype=e.rdt)
        full = e.grf_hc(99, 7, cs)
        for (w, rb) in ((37, 20), (0, 9), (64, 0), (300, 200)):
            out = e.hc(); out[:] = 5.0
            e.grf_hc(99, 7, cs, out=out, width=w, rband=rb)
            rows = np.r_[0:rb, 256 - rb + 1:256] if (rb and 2 * rb - 1 < 256) else np.arange(256)
            wv = ((w + 1) // 2) * 2 if (0 < w < 257 and (w + 1) // 2 < 129) else 257      # whole column pairs are drawn
            assert torch.equal(out[rows][:, :min(wv, 257)], full[rows][:, :min(wv, 257)])
            other = np.setdiff1d(np.arange(256), rows)
            assert bool((out[other] == 5.0).all()) and bool((out[:, min(wv, 257):257] == 5.0).all())


def test_grf_statistics():
    ny, nx = 256, 256
    e = eng(ny, nx, "f32")
    k = e.grf_hc(1234, 0)
    m = e.irfft(k, scale=1.0 / np.sqrt(ny * nx)).cpu().numpy()
    assert abs(m.mean()) < 5.0 / np.sqrt(ny * nx) * 3
    assert abs(m.var() - 1.0) < 0.03
    # independent stream ids differ, same id reproduces
    k2 = e.grf_hc(1234, 1); k3 = e.grf_hc(1234, 0)
    assert not torch.equal(k, k2) and torch.equal(k, k3)
    # Hermitian self-consistency of the kx=0 column
    kk = k.cpu().numpy()
    assert np.allclose(kk[1:ny // 2, 0], np.conj(kk[:ny // 2:-1, 0]))
    assert kk[0, 0].imag == 0 and kk[ny // 2, nx // 2].imag == 0
    r = e.randn(7, 3).cpu().numpy()
    assert abs(r.mean()) < 0.02 and abs(r.std() - 1) < 0.02


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_bin_power_equals_f2power_then_bin(prec):
    from orphics_amd.geometry import FlatGeometry
    ny, nx = 128, 256
    e = eng(ny, nx, prec)
    g = FlatGeometry.from_res((ny, nx), 2.0)
    e.set_laxes(*g.laxes())
    ed = torch.as_tensor(np.arange(100., 3000., 200.), device=e.device)
    rng = np.random.default_rng(3)
    k1 = e.rfft(e.to_real(rng.standard_normal((ny, nx))))
    k2 = e.rfft(e.to_real(rng.standard_normal((ny, nx))))
    ids = e.modl_digitize(ed, half=True)
    nids = ed.numel() + 1
    for a, b in ((k1, k1), (k1, k2)):
        s1, c1 = e.bin(e.f2power(a, b, 0.37), ids, nids, herm=True)
        s2, c2 = e.bin_power(a, b, 0.37, ids, nids, herm=True)
        assert torch.equal(c1, c2)
        np.testing.assert_allclose(s2.cpu().numpy(), s1.cpu().numpy(), rtol=1e-12 if prec == "f64" else 1e-6)


def test_error_behaviour_is_loud():
    """Bad shapes / arguments are rejected on the host before any kernel is launched; the C-ABI returns non-zero
    with oa_last_error() set and never throws."""
    import ctypes
    from orphics_amd import _lib, maps, stats
    from orphics_amd.geometry import FlatGeometry
    lib = _lib.load()
    e = eng(64, 64, "f32")
    with pytest.raises(ValueError):
        e.rfft(torch.zeros(32, 64, device="cuda"))                      # wrong shape
    with pytest.raises(ValueError):
        e.rfft(torch.zeros(64, 64, device="cuda", dtype=torch.float64))  # wrong dtype for an f32 plan
    with pytest.raises(TypeError):
        e.rfft(np.zeros((64, 64), np.float32))                           # host memory is not a device plane
    with pytest.raises(NotImplementedError):
        maps.FourierCalc((101, 100), FlatGeometry.from_res((101, 100), 2.0)).fft(np.zeros((101, 100)))   # odd side
    with pytest.raises(ValueError):
        stats.bin2D(np.ones((8, 8)), np.array([3., 2., 1.]))
    with pytest.raises(TypeError):
        stats.bin2D(np.ones((8, 8)), np.array([1., 2., 3.])).bin(np.ones((8, 8)) * 1j)
    h = ctypes.c_void_p()
    assert lib.oa_plan_create(101, 64, 0, ctypes.byref(h)) != 0 and b"must be even" in lib.oa_last_error()
    assert lib.oa_plan_create(10000, 64, 0, ctypes.byref(h)) != 0 and b"<= 8192" in lib.oa_last_error()
    assert lib.oa_plan_create(16, 64, 0, ctypes.byref(h)) != 0 and b">= 32" in lib.oa_last_error()
    assert lib.oa_plan_create(64, 64, 7, ctypes.byref(h)) != 0 and b"dtype" in lib.oa_last_error()
    assert lib.oa_fft_r2c(e.plan, None, None, 1.0, 0, 0, None) != 0 and b"NULL" in lib.oa_last_error()
    x = torch.zeros(64, 64, device="cuda")
    assert lib.oa_bin(0, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(x.data_ptr()), None, None, 10, 5000, 0, 0, 0, -1,
                      ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(x.data_ptr()), None, ctypes.c_void_p(x.data_ptr()), None) != 0
    assert b"nids" in lib.oa_last_error()
    # legs need the multipole axes first
    e2 = __import__("orphics_amd.engine", fromlist=["Engine"]).Engine(64, 64, "f32")
    with pytest.raises(_lib.OrphicsAmdError):
        e2.qe_legs(e2.hc(), e2.hc(), e2.hcreal(), e2.hcreal())


@pytest.mark.parametrize("ny,nx", [(96, 160), (600, 750), (250, 36), (1200, 1200), (112, 154), (66, 98)])
def test_non_power_of_two_sides(ny, nx):
    """Even sides that are not powers of two: 2^a 3^b 5^c (the reference notebooks' 600, 750, 1200, 2400) are mixed-radix transforms
    (csrc/fft_mixed.hpp), sides with another prime factor (112 = 2^4 7, 154 = 2 7 11, 66, 98) the chirp-z path (csrc/czt.hip):
    rfft / irfft / cfft equal NumPy; the fused estimator entry points refuse loudly."""
    from orphics_amd.engine import Engine
    from orphics_amd._lib import OrphicsAmdError
    rng = np.random.default_rng(ny + nx)
    x = rng.standard_normal((ny, nx))
    z = rng.standard_normal((ny, nx)) + 1j * rng.standard_normal((ny, nx))
    for prec, tol in (("f64", 1e-11), ("f32", 2e-5)):
        e = Engine(ny, nx, prec)
        assert not e.pow2
        xt = torch.as_tensor(x, dtype=e.rdt, device=e.device)
        k = e.rfft(xt)
        ref = np.fft.rfft2(x)
        assert np.abs(k.cpu().numpy()[:, :nx // 2 + 1] - ref).max() < tol * np.abs(ref).max()
        back = e.irfft(k)
        assert np.abs(back.cpu().numpy() - x).max() < tol * 10
        zt = torch.as_tensor(z, dtype=e.cdt, device=e.device)
        f = e.cfft(zt)
        reff = np.fft.fft2(z)
        assert np.abs(f.cpu().numpy() - reff).max() < tol * np.abs(reff).max()
        b = e.cfft(f, inverse=True, scale=1.0 / (ny * nx))
        assert np.abs(b.cpu().numpy() - z).max() < tol * 10
        with pytest.raises(OrphicsAmdError, match="power-of-two"):
            e.fft_cols(k)
