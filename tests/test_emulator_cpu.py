"""The FFT / fused-estimator pass bodies (orphics_amd/csrc/fft_kernels.hpp) run under the CPU thread
emulator (tests/emul: std::thread + std::barrier, one std::thread per HIP thread) and are compared with
NumPy.  This validates every index computation of the kernels -- including the active-column (pruned)
variants -- without a GPU; the GPU tests then only have to confirm the same code on the device."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EMUL = os.path.join(HERE, "emul")


@pytest.fixture(scope="module")
def emu():
    so = os.path.join(EMUL, "libemul_fft.so")
    src = os.path.join(EMUL, "emul_fft.cpp")
    csrc = os.path.join(HERE, "..", "orphics_amd", "csrc")
    hdrs = [os.path.join(csrc, h) for h in ("fft_kernels.hpp", "fft_plan.hpp", "fft_r2c_w64.hpp", "fft_r2c_rs4096.hpp", "fft_fband.hpp", "fft_rowqe8.hpp", "fft_mixed.hpp", "cx.hpp")]
    if (not os.path.exists(so)) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++20", "-fPIC", "-shared", "-pthread", "-o", so, src])
    lib = ctypes.CDLL(so)
    lib.emu_kpitch.restype = ctypes.c_long
    return lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _hc(lib, ny, nx, fill=0.0):
    kp = lib.emu_kpitch(nx)
    a = np.zeros((ny, kp), dtype=np.complex128)
    a[:] = fill
    return a


def _band_limited(rng, ny, nx, w):
    """Random Hermitian-consistent hc plane that vanishes for columns >= w (and is real at kx = 0 self-pairs)."""
    x = rng.standard_normal((ny, nx))
    k = np.fft.rfft2(x)
    k[:, w:] = 0
    return k


@pytest.mark.parametrize("ny,nx", [(32, 32), (64, 128), (128, 64), (32, 512)])
def test_r2c_c2r_match_numpy(emu, ny, nx):
    rng = np.random.default_rng(ny * 1000 + nx)
    x = rng.standard_normal((ny, nx))
    out = _hc(emu, ny, nx)
    assert emu.emu_r2c_f64(ny, nx, _p(x), _p(out), ctypes.c_double(1.0)) == 0
    ref = np.fft.rfft2(x)
    assert np.abs(out[:, :nx // 2 + 1] - ref).max() < 1e-11 * np.abs(ref).max()
    back = np.zeros((ny, nx))
    assert emu.emu_c2r_f64(ny, nx, _p(out), _p(back), ctypes.c_double(1.0 / (ny * nx))) == 0
    assert np.abs(back - x).max() < 1e-12
    # float32 build of the same bodies
    x32 = x.astype(np.float32)
    kp = emu.emu_kpitch(nx)
    o32 = np.zeros((ny, kp), dtype=np.complex64)
    assert emu.emu_r2c_f32(ny, nx, _p(x32), _p(o32), ctypes.c_double(1.0)) == 0
    assert np.abs(o32[:, :nx // 2 + 1] - ref).max() < 2e-5 * np.abs(ref).max()


@pytest.mark.parametrize("ny,nx,w", [(64, 128, 9), (64, 128, 32), (32, 512, 70), (128, 64, 33)])
def test_active_columns_r2c_c2r(emu, ny, nx, w):
    """width-limited R2C writes exactly the leading columns (same values) and nothing else; width-limited C2R
    of a band-limited plane ignores whatever sits beyond the band."""
    rng = np.random.default_rng(7 + w)
    x = rng.standard_normal((ny, nx))
    full = _hc(emu, ny, nx)
    emu.emu_r2c_f64(ny, nx, _p(x), _p(full), ctypes.c_double(1.0))
    part = _hc(emu, ny, nx, fill=99.0)
    assert emu.emu_r2c_w_f64(ny, nx, _p(x), _p(part), ctypes.c_double(1.0), w, 0) == 0
    wv = min(w, nx // 2 + 1)
    assert np.array_equal(part[:, :wv], full[:, :wv])
    assert np.all(part[:, wv:] == 99.0)
    # + row band: only rows y < rb or y > ny - rb of the leading columns are produced
    rb = max(2, ny // 8)
    band = np.r_[0:rb, ny - rb + 1:ny]
    off = np.r_[rb:ny - rb + 1]
    part2 = _hc(emu, ny, nx, fill=99.0)
    assert emu.emu_r2c_w_f64(ny, nx, _p(x), _p(part2), ctypes.c_double(1.0), w, rb) == 0
    assert np.array_equal(part2[band][:, :wv], full[band][:, :wv])
    assert np.all(part2[:, wv:] == 99.0)                  # rows outside the band hold pass-1 intermediates: undefined
    k = _hc(emu, ny, nx)
    k[:, :nx // 2 + 1] = _band_limited(rng, ny, nx, wv)
    ref = np.fft.irfft2(k[:, :nx // 2 + 1], s=(ny, nx))
    k[:, wv:] = 1e30                                      # must never be read
    back = np.zeros((ny, nx))
    assert emu.emu_c2r_w_f64(ny, nx, _p(k), _p(back), ctypes.c_double(1.0 / (ny * nx)), w) == 0
    assert np.abs(back - ref).max() < 1e-12 * max(1.0, np.abs(ref).max())


def _col_ifft(k, ny):
    return np.fft.ifft(k, axis=0) * ny


@pytest.mark.parametrize("stockham", [0, 1])
@pytest.mark.parametrize("ny,nx,win,wout", [(32, 64, 0, 0), (32, 128, 20, 41), (64, 256, 33, 64), (32, 512, 100, 150),
                                            (32, 2048, 200, 330), (32, 8192, 380, 664), (32, 8192, 200, 330),
                                            (64, 512, 30, 60), (32, 16384, 380, 664), (32, 16384, 0, 0)])
def test_fused_row_stage(emu, ny, nx, win, wout, stockham):
    """row_qe: P = R2C(C2R(G) * C2R(H)) row by row, with and without active-column limits; both kernel bodies
    (Stockham = the product kernel, in-place DIF/DIT = the -DOA_QE_INPLACE variant with 18 instead of 30 barriers).
    The 8192 and 512 cases with small win run the active-column first stage (NZ = 2, 1, 2 live taps per side).
    full8: 8192-point rows through the two-rows-per-transform stage with the radix-16 cross stage (row_qe8_body, A = 16: what
    Fft2dPlan::rows_qe launches for the map's own row length at 8192^2)."""
    _run_fused_row_stage(emu, ny, nx, win, wout, stockham, False)


@pytest.mark.parametrize("ny,win,wout", [(8, 380, 664), (4, 1139, 664), (4, 2048, 4097)])
def test_fused_row_stage_on_8192_point_rows_two_rows_per_transform(emu, ny, win, wout):
    """The map's own row length at 8192^2 (row_grid = "full" and the dense pipeline): row_qe8_body with A = 16 -- two threads per
    position of the radix-16 cross-wave stage, the Nyquist column taken once, products on the grid the reference forms them on (no
    alias argument) -- against NumPy, with 1 and 4 live 512-point blocks per side (wider bands keep the packed kernel)."""
    _run_fused_row_stage(emu, ny, 8192, win, wout, 1, True)


def _run_fused_row_stage(emu, ny, nx, win, wout, stockham, full8):
    emu.emu_set_stockham_qe(stockham)
    emu.emu_set_rowqe8(1 if full8 else 0)
    rng = np.random.default_rng(100 + nx + win)
    W = nx // 2 + 1
    wi = win if win else W
    planes = []
    for _ in range(3):      # inputs are column-transformed legs: Hermitian along x only matters per row
        k = (rng.standard_normal((ny, W)) + 1j * rng.standard_normal((ny, W)))
        k[:, 0] = k[:, 0].real
        if nx // 2 < wi:
            k[:, nx // 2] = k[:, nx // 2].real
        k[:, wi:] = 0
        planes.append(k)
    gx, gy, h = planes
    rows = lambda k: np.fft.irfft(k, n=nx, axis=1) * nx          # unnormalised C2R rows
    hr = rows(h)
    ref = [np.fft.rfft(rows(g) * hr, axis=1) for g in (gx, gy)]
    ins = []
    for k in planes:
        a = _hc(emu, ny, nx, fill=(1e30 if win else 0.0))      # garbage beyond the band must not be read
        a[:, :wi] = k[:, :wi]
        ins.append(a)
    px, py = _hc(emu, ny, nx, fill=5.0), _hc(emu, ny, nx, fill=5.0)
    assert emu.emu_qe_rows_w_f64(ny, nx, _p(ins[0]), _p(ins[1]), _p(ins[2]), _p(px), _p(py), ctypes.c_double(1.0), win, wout) == 0
    wo = wout if wout else W
    for got, want in ((px, ref[0]), (py, ref[1])):
        assert np.abs(got[:, :wo] - want[:, :wo]).max() < 1e-11 * np.abs(want).max()
        if wout:
            assert np.all(got[:, wo:W] == 5.0)                    # untouched
    emu.emu_set_rowqe8(1)


@pytest.fixture(params=[8, 16], ids=["8pt", "16pt"])
def body(emu, request):
    """which two-rows-per-transform row stage Fft2dPlan::rows_qe launches: 8 points per thread (fft_rowqe8.hpp, the default; also the
    only one with the 3 x 512-point grid) or 16 (row_qe_pair_body)"""
    emu.emu_set_rowqe8(1 if request.param == 8 else 0)
    yield request.param
    emu.emu_set_rowqe8(1)


@pytest.mark.parametrize("ny,nx,win,wout,mrow,expect", [(32, 8192, 380, 664, -1, (1536, 2048)), (32, 8192, 1139, 664, -1, 4096),
                                                         (32, 2048, 100, 150, -1, 1024), (32, 1024, 60, 100, 512, 512),
                                                         (32, 512, 100, 150, -1, 512), (32, 4096, 190, 332, -1, 1024),
                                                         (16, 16384, 380, 664, -1, (1536, 2048)), (8, 8192, 380, 664, 2048, 2048),
                                                         (8, 4096, 512, 512, -1, (1536, 2048)), (8, 8192, 513, 400, -1, 2048)])
def test_fused_row_stage_on_alias_free_row_grid(emu, body, ny, nx, win, wout, mrow, expect):
    """ROW GRID (include/orphics_amd.h): band-limited legs (columns >= win vanish) -> the row stage on a grid of
    mrow >= 2 win + wout points returns the same product columns k < wout as the full-length transform
    (through the actual kernel body, here at a shorter compile-time row length).  Both bodies; the 8-point one also on
    the 1536-point grid (radix-3 cross-wave stage)."""
    if isinstance(expect, tuple):
        expect = expect[0] if body == 8 else expect[1]
    emu.emu_set_stockham_qe(1)
    rng = np.random.default_rng(7 + nx + win)
    W = nx // 2 + 1
    planes = []
    for _ in range(3):
        k = (rng.standard_normal((ny, W)) + 1j * rng.standard_normal((ny, W)))
        k[:, 0] = k[:, 0].real
        k[:, win:] = 0
        planes.append(k)
    gx, gy, h = planes
    rows = lambda k: np.fft.irfft(k, n=nx, axis=1) * nx
    hr = rows(h)
    ref = [np.fft.rfft(rows(g) * hr, axis=1) for g in (gx, gy)]
    ins = []
    for k in planes:
        a = _hc(emu, ny, nx, fill=1e30)                         # garbage beyond the band must not be read
        a[:, :win] = k[:, :win]
        ins.append(a)
    px, py = _hc(emu, ny, nx, fill=5.0), _hc(emu, ny, nx, fill=5.0)
    used = emu.emu_qe_rows_wm_f64(ny, nx, _p(ins[0]), _p(ins[1]), _p(ins[2]), _p(px), _p(py), ctypes.c_double(1.0), win, wout, mrow)
    assert used == expect
    for got, want in ((px, ref[0]), (py, ref[1])):
        assert np.abs(got[:, :wout] - want[:, :wout]).max() < 1e-11 * np.abs(want).max()
        assert np.all(got[:, wout:W] == 5.0)


@pytest.mark.parametrize("mrow", [1024, 1536])
def test_row_stage_of_several_maps_per_launch(emu, body, mrow):
    """row_qe_pair_body with workgroup ranges = maps: evenly spaced planes (oa_mc_run, oa_qe_tt_moments2; the h planes may
    have their own spacing) and a table of per-map planes and scales (oa_qe_mv) give, map by map, bit for bit what one
    launch per map gives -- also when accumulating."""
    if body == 16 and mrow == 1536:
        pytest.skip("the 3 x 512-point grid exists in the 8-point body only")
    ny, nx, win, wout = 8, 4096, 100, 150
    rng = np.random.default_rng(77)
    kp = emu.emu_kpitch(nx)
    nm = 3
    G = np.full((2 * nm, ny, kp), 1e30 + 0j)           # gx_m, gy_m interleaved: even spacing 2 planes
    H = np.full((nm, ny, kp), 1e30 + 0j)               # h_m: spacing 1 plane
    for a in (G, H):
        a[..., :win] = rng.standard_normal(a.shape[:-1] + (win,)) + 1j * rng.standard_normal(a.shape[:-1] + (win,))
        a[..., 0] = a[..., 0].real
    scales = np.array([1.0, -0.5, 2.0])
    arr = lambda xs: (ctypes.c_void_p * len(xs))(*[x.ctypes.data for x in xs])      # noqa: E731
    one = np.zeros((2 * nm, ny, kp), dtype=np.complex128)
    for m in range(nm):                                  # reference: one launch per map, twice (the second accumulates)
        for acc in (0, 1):
            sc = (ctypes.c_double * 1)(scales[m])
            assert emu.emu_qe_rows_multi_f64(ny, nx, 1, arr([G[2 * m]]), arr([G[2 * m + 1]]), arr([H[m]]), arr([one[2 * m]]), arr([one[2 * m + 1]]),
                                             sc, acc, win, wout, mrow, 0) == 0
    assert np.abs(one[:, :, :wout]).min() > 0
    # evenly spaced: one scale for all maps
    even = np.zeros_like(one)
    ref_even = np.zeros_like(one)
    sc1 = (ctypes.c_double * 1)(0.75)
    for m in range(nm):
        assert emu.emu_qe_rows_multi_f64(ny, nx, 1, arr([G[2 * m]]), arr([G[2 * m + 1]]), arr([H[m]]), arr([ref_even[2 * m]]), arr([ref_even[2 * m + 1]]),
                                         sc1, 0, win, wout, mrow, 0) == 0
    sc3 = (ctypes.c_double * nm)(0.75, 0.75, 0.75)
    assert emu.emu_qe_rows_multi_f64(ny, nx, nm, arr([G[2 * m] for m in range(nm)]), arr([G[2 * m + 1] for m in range(nm)]), arr([H[m] for m in range(nm)]),
                                     arr([even[2 * m] for m in range(nm)]), arr([even[2 * m + 1] for m in range(nm)]), sc3, 0, win, wout, mrow, 0) == 0
    assert np.array_equal(even, ref_even)
    # table: arbitrary plane order (maps reversed), per-map scales, two launches (the second accumulates)
    tab = np.zeros_like(one)
    order = [2, 0, 1]
    scs = (ctypes.c_double * nm)(*[scales[m] for m in order])
    for acc in (0, 1):
        assert emu.emu_qe_rows_multi_f64(ny, nx, nm, arr([G[2 * m] for m in order]), arr([G[2 * m + 1] for m in order]), arr([H[m] for m in order]),
                                         arr([tab[2 * m] for m in order]), arr([tab[2 * m + 1] for m in order]), scs, acc, win, wout, mrow, 1) == 0
    assert np.array_equal(tab, one)


@pytest.mark.parametrize("mrow", [1024, 1536])
def test_row_stage_estimator_chains(emu, body, mrow):
    """row_qe_pair_body<.., CHAIN>: every estimator's pieces in ONE launch -- per piece three inverse transforms and the real-space
    product, summed over the pieces in registers, one forward pair per estimator -- against piece-by-piece launches that
    accumulate in the product planes (the linear forward transform commutes with the sum)."""
    if body == 16 and mrow == 1536:
        pytest.skip("the 3 x 512-point grid exists in the 8-point body only")
    ny, nx, win, wout = 8, 4096, 100, 150
    rng = np.random.default_rng(78)
    kp = emu.emu_kpitch(nx)
    counts = [1, 3, 2]
    total = sum(counts)
    G = np.full((2 * total, ny, kp), 1e30 + 0j)
    H = np.full((total, ny, kp), 1e30 + 0j)
    for a in (G, H):
        a[..., :win] = rng.standard_normal(a.shape[:-1] + (win,)) + 1j * rng.standard_normal(a.shape[:-1] + (win,))
        a[..., 0] = a[..., 0].real
    scales = rng.uniform(-1.5, 1.5, total)
    arr = lambda xs: (ctypes.c_void_p * len(xs))(*[x.ctypes.data for x in xs])      # noqa: E731
    ref = np.zeros((2 * len(counts), ny, kp), dtype=np.complex128)
    i = 0
    for e, n in enumerate(counts):
        for k in range(n):
            sc = (ctypes.c_double * 1)(scales[i])
            assert emu.emu_qe_rows_multi_f64(ny, nx, 1, arr([G[2 * i]]), arr([G[2 * i + 1]]), arr([H[i]]), arr([ref[2 * e]]), arr([ref[2 * e + 1]]),
                                             sc, 1 if k else 0, win, wout, mrow, 0) == 0
            i += 1
    got = np.full_like(ref, 9.0)
    owner = [e for e, n in enumerate(counts) for _ in range(n)]
    first = (ctypes.c_int * len(counts))(*np.cumsum([0] + counts[:-1]).tolist())
    cnt = (ctypes.c_int * len(counts))(*counts)
    assert emu.emu_qe_rows_chain_f64(ny, nx, len(counts), total, arr([G[2 * i] for i in range(total)]), arr([G[2 * i + 1] for i in range(total)]),
                                     arr([H[i] for i in range(total)]), arr([got[2 * owner[i]] for i in range(total)]),
                                     arr([got[2 * owner[i] + 1] for i in range(total)]), (ctypes.c_double * total)(*scales), first, cnt, win, wout, mrow) == 0
    assert np.abs(got[:, :, :wout] - ref[:, :, :wout]).max() < 1e-12 * np.abs(ref[:, :, :wout]).max()
    assert np.all(got[:, :, wout:] == 9.0)


@pytest.mark.parametrize("ny,nx,w,rb", [(64, 64, 0, 0), (64, 128, 21, 0), (128, 64, 32, 9), (256, 64, 7, 40), (64, 64, 0, 5)])
def test_fused_column_stages(emu, ny, nx, w, rb):
    """col_legs (+ pass 2) = inverse column transforms of (i lx FG kX, i ly FG kX, FH kY);
    col_div (after 2 pass-1 launches) = Fn * (i lx FFTcol[A] + i ly FFTcol[B]); both with active widths."""
    rng = np.random.default_rng(5 + ny + w)
    W = nx // 2 + 1
    wv = w if w else W
    kp = emu.emu_kpitch(nx)
    ly = 2 * np.pi * np.fft.fftfreq(ny) * 100
    lx = 2 * np.pi * np.fft.fftfreq(nx) * 100
    lyd, lxd = ly.copy(), lx.copy()
    lyd[ny // 2] = 0
    lxd[nx // 2] = 0
    kX = _hc(emu, ny, nx); kY = _hc(emu, ny, nx)
    kX[:, :W] = rng.standard_normal((ny, W)) + 1j * rng.standard_normal((ny, W))
    kY[:, :W] = rng.standard_normal((ny, W)) + 1j * rng.standard_normal((ny, W))
    FG = np.zeros((ny, kp)); FH = np.zeros((ny, kp)); Fn = np.zeros((ny, kp))
    FG[:, :wv] = rng.uniform(0.5, 1.5, (ny, wv))
    FH[:, :wv] = rng.uniform(0.5, 1.5, (ny, wv))
    Fn[:, :wv] = rng.uniform(0.5, 1.5, (ny, wv))
    off = np.r_[rb:ny - rb + 1] if rb else np.zeros(0, dtype=int)
    FG[off] = 0; FH[off] = 0; Fn[off] = 0                # filters vanish outside the row band ...
    kXg, kYg = kX.copy(), kY.copy()
    kXg[off] = 1e30; kYg[off] = 1e30                     # ... where the inputs must never be read
    outs = [_hc(emu, ny, nx, fill=3.0) for _ in range(3)]
    lxh = np.ascontiguousarray(lxd)                     # the kernel indexes lxd by column (first W entries used)
    assert emu.emu_legs_cols_w_f64(ny, nx, _p(kXg), _p(kYg), _p(FG), _p(FH), _p(lxh), _p(lyd), _p(outs[0]), _p(outs[1]),
                                   _p(outs[2]), w, rb) == 0
    lx2, ly2 = lxd[None, :W], lyd[:, None]
    refs = [_col_ifft(1j * lx2 * FG[:, :W] * kX[:, :W], ny), _col_ifft(1j * ly2 * FG[:, :W] * kX[:, :W], ny),
            _col_ifft(FH[:, :W] * kY[:, :W], ny)]
    for got, want in zip(outs, refs):
        assert np.abs(got[:, :wv] - want[:, :wv]).max() < 1e-11 * np.abs(want).max()
        if w:
            assert np.all(got[:, wv:W] == 3.0)
    A = _hc(emu, ny, nx); B = _hc(emu, ny, nx)
    A[:, :W] = rng.standard_normal((ny, W)) + 1j * rng.standard_normal((ny, W))
    B[:, :W] = rng.standard_normal((ny, W)) + 1j * rng.standard_normal((ny, W))
    out = _hc(emu, ny, nx, fill=3.0)
    assert emu.emu_cols_div_w_f64(ny, nx, _p(A), _p(B), _p(Fn), _p(lxh), _p(lyd), _p(out), w, rb) == 0
    want = Fn[:, :W] * (1j * lx2 * np.fft.fft(A[:, :W], axis=0) + 1j * ly2 * np.fft.fft(B[:, :W], axis=0))
    on = np.setdiff1d(np.arange(ny), off)
    assert np.abs(out[on][:, :wv] - want[on][:, :wv]).max() < 1e-11 * np.abs(want).max()
    if w:
        assert np.all(out[:, wv:W] == 3.0)
    if rb:
        assert np.all(out[off] == 3.0)                    # rows outside the band are not written


@pytest.mark.parametrize("ny,nx,w,rb", [(1024, 64, 0, 0), (2048, 64, 20, 100), (8192, 64, 33, 300)])
def test_fused_forward_pass2_legs(emu, ny, nx, w, rb):
    """col_fwdlegs: real map -> (row R2C, forward column pass 1) -> ONE kernel doing forward pass 2 + leg filters +
    inverse pass 1 with the column length split the other way round -> inverse pass 2.  Must equal the inverse column
    transforms of (i lx FG kT, i ly FG kT, FH kT) with kT = rfft2(map); 2048 / 8192 exercise the asymmetric split."""
    rng = np.random.default_rng(ny + w)
    W = nx // 2 + 1
    wv = w if w else W
    kp = emu.emu_kpitch(nx)
    x = rng.standard_normal((ny, nx))
    kT = np.fft.rfft2(x)
    lyd = 2 * np.pi * np.fft.fftfreq(ny) * 100
    lxd = 2 * np.pi * np.fft.fftfreq(nx) * 100
    lyd[ny // 2] = 0
    lxd[nx // 2] = 0
    FG = np.zeros((ny, kp)); FH = np.zeros((ny, kp))
    FG[:, :wv] = rng.uniform(0.5, 1.5, (ny, wv))
    FH[:, :wv] = rng.uniform(0.5, 1.5, (ny, wv))
    if rb:
        off = np.r_[rb:ny - rb + 1]
        FG[off] = 0; FH[off] = 0
    outs = [_hc(emu, ny, nx, fill=3.0) for _ in range(3)]
    assert emu.emu_map_legs_cols_f64(ny, nx, _p(x), _p(FG), _p(FH), _p(lxd), _p(lyd), _p(outs[0]), _p(outs[1]), _p(outs[2]),
                                     w, rb) == 0
    lx2, ly2 = lxd[None, :W], lyd[:, None]
    refs = [_col_ifft(1j * lx2 * FG[:, :W] * kT, ny), _col_ifft(1j * ly2 * FG[:, :W] * kT, ny), _col_ifft(FH[:, :W] * kT, ny)]
    for got, want in zip(outs, refs):
        assert np.abs(got[:, :wv] - want[:, :wv]).max() < 1e-10 * np.abs(want).max()
        if w:
            assert np.all(got[:, wv:W] == 3.0)


@pytest.mark.parametrize("ny_full,my,w,rb", [(256, 64, 21, 9), (512, 128, 0, 20), (256, 128, 30, 33)])
def test_column_grid_views(emu, ny_full, my, w, rb):
    """COLUMN GRID (include/orphics_amd.h): col_legs / col_div run on my < ny rows while filters, the ly axis, kX / kY
    and the output keep the full-resolution row layout -- row y of the my-row transform is row
    y + (y >= my/2 ? ny - my : 0) of the full planes."""
    nx = 64
    rng = np.random.default_rng(ny_full + my + w)
    W = nx // 2 + 1
    wv = w if w else W
    kp = emu.emu_kpitch(nx)
    ly = 2 * np.pi * np.fft.fftfreq(ny_full) * 100
    lx = 2 * np.pi * np.fft.fftfreq(nx) * 100
    lyd, lxd = ly.copy(), lx.copy()
    lyd[ny_full // 2] = 0
    lxd[nx // 2] = 0
    rows = np.r_[0:my // 2, ny_full - my // 2:ny_full]            # full-resolution row of each coarse row
    band = np.r_[0:rb, ny_full - rb + 1:ny_full]
    kX = _hc(emu, ny_full, nx, fill=1e30); kY = _hc(emu, ny_full, nx, fill=1e30)    # garbage wherever it must not be read
    kX[band, :wv] = rng.standard_normal((band.size, wv)) + 1j * rng.standard_normal((band.size, wv))
    kY[band, :wv] = rng.standard_normal((band.size, wv)) + 1j * rng.standard_normal((band.size, wv))
    FG = np.zeros((ny_full, kp)); FH = np.zeros((ny_full, kp)); Fn = np.zeros((ny_full, kp))
    FG[band, :wv] = rng.uniform(0.5, 1.5, (band.size, wv))
    FH[band, :wv] = rng.uniform(0.5, 1.5, (band.size, wv))
    Fn[band, :wv] = rng.uniform(0.5, 1.5, (band.size, wv))
    outs = [_hc(emu, my, nx, fill=3.0) for _ in range(3)]
    assert emu.emu_legs_cols_cg_f64(ny_full, my, nx, _p(kX), _p(kY), _p(FG), _p(FH), _p(lxd), _p(lyd), _p(outs[0]), _p(outs[1]),
                                    _p(outs[2]), w, rb) == 0
    kXc = np.where(FG[rows][:, :W] != 0, kX[rows][:, :W], 0)      # the coarse spectrum: band rows, zero elsewhere
    kYc = np.where(FH[rows][:, :W] != 0, kY[rows][:, :W], 0)
    lx2, ly2 = lxd[None, :W], lyd[rows][:, None]
    refs = [_col_ifft(1j * lx2 * FG[rows][:, :W] * kXc, my), _col_ifft(1j * ly2 * FG[rows][:, :W] * kXc, my),
            _col_ifft(FH[rows][:, :W] * kYc, my)]
    for got, want in zip(outs, refs):
        assert np.abs(got[:, :wv] - want[:, :wv]).max() < 1e-11 * np.abs(want).max()
        if w:
            assert np.all(got[:, wv:W] == 3.0)
    A = _hc(emu, my, nx); B = _hc(emu, my, nx)
    A[:, :W] = rng.standard_normal((my, W)) + 1j * rng.standard_normal((my, W))
    B[:, :W] = rng.standard_normal((my, W)) + 1j * rng.standard_normal((my, W))
    out = _hc(emu, ny_full, nx, fill=3.0)
    assert emu.emu_cols_div_cg_f64(ny_full, my, nx, _p(A), _p(B), _p(Fn), _p(lxd), _p(lyd), _p(out), w, rb) == 0
    want = Fn[rows][:, :W] * (1j * lx2 * np.fft.fft(A[:, :W], axis=0) + 1j * ly2 * np.fft.fft(B[:, :W], axis=0))
    coarse_band = np.r_[0:rb, my - rb + 1:my]
    assert np.abs(out[rows[coarse_band]][:, :wv] - want[coarse_band][:, :wv]).max() < 1e-11 * np.abs(want).max()
    untouched = np.setdiff1d(np.arange(ny_full), rows[coarse_band])
    assert np.all(out[untouched] == 3.0)                         # nothing outside kappa's band rows is written
    if w:
        assert np.all(out[:, wv:W] == 3.0)


@pytest.mark.parametrize("ny_full,my,w,rb", [(256, 64, 21, 9), (256, 128, 0, 33), (4096, 1024, 11, 200)])
def test_batched_leg_and_divergence_launches(emu, ny_full, my, w, rb):
    """oa_qe_mv's launches (include/orphics_amd.h): every distinct filtered field of several estimators in ONE col_legs launch
    (grid z = leg plane, source by a 2-bit field, filter planes through a pointer table) + one pass-2 launch over the pool;
    the divergence of several estimators in one launch (grid z, per-estimator Fn / product / output offsets).  Same results
    as the one-field / one-estimator launches."""
    nx = 64
    rng = np.random.default_rng(ny_full + my)
    W = nx // 2 + 1
    wv = w if w else W
    kp = emu.emu_kpitch(nx)
    ly = 2 * np.pi * np.fft.fftfreq(ny_full) * 100
    lx = 2 * np.pi * np.fft.fftfreq(nx) * 100
    lyd, lxd = ly.copy(), lx.copy()
    lyd[ny_full // 2] = 0
    lxd[nx // 2] = 0
    band = np.r_[0:rb, ny_full - rb + 1:ny_full]
    src = np.full((3, ny_full, kp), 1e30 + 0j)                       # three sources (T, E, B) in one block
    src[:, band, :wv] = rng.standard_normal((3, band.size, wv)) + 1j * rng.standard_normal((3, band.size, wv))
    ngrad, nh = 3, 2
    filt = np.zeros((ngrad + nh, ny_full, kp))
    filt[:, band, :wv] = rng.uniform(0.5, 1.5, (ngrad + nh, band.size, wv))
    which = [0, 2, 1, 1, 0]                                          # source of each field (gradient fields first)
    srcsel = sum(k << (2 * f) for f, k in enumerate(which))
    ftab = (ctypes.c_void_p * (ngrad + nh))(*[filt[f].ctypes.data for f in range(ngrad + nh)])
    nplanes = 2 * ngrad + nh
    pool = np.full((nplanes, my, kp), 3.0 + 0j)
    plane = ny_full * kp
    assert emu.emu_legs_batch_cg_f64(ny_full, my, nx, _p(src), ctypes.c_long(plane), ctypes.c_long(2 * plane), ctypes.c_ulonglong(srcsel),
                                     ftab, ngrad, nh, _p(lxd), _p(lyd), _p(pool), ctypes.c_long(my * kp), w, rb) == 0
    for f in range(ngrad + nh):
        ref = [_hc(emu, my, nx, fill=3.0) for _ in range(3)]
        k = np.ascontiguousarray(src[which[f]]); F = np.ascontiguousarray(filt[f])
        assert emu.emu_legs_cols_cg_f64(ny_full, my, nx, _p(k), _p(k), _p(F), _p(F), _p(lxd), _p(lyd), _p(ref[0]), _p(ref[1]), _p(ref[2]), w, rb) == 0
        if my >= 1024:
            # single-pass leg kernel (col_legs_sp: a whole 1024-point column in the tile) against the two-pass launches: another
            # factorisation of the same transform -> rounding; nothing outside the kept columns is written
            close = lambda a_, b_: np.abs(a_ - b_).max() <= 1e-12 * np.abs(b_).max()      # noqa: E731
        else:
            close = np.array_equal
        if f < ngrad:
            assert close(pool[2 * f], ref[0]) and close(pool[2 * f + 1], ref[1])
        else:
            assert close(pool[ngrad + f], ref[2])
    if my >= 1024:
        assert np.all(pool[:, :, wv:] == 3.0)
        return
    # divergence of three estimators in one launch
    ne = 3
    for dt, cdt, fn in ((np.float64, np.complex128, emu.emu_cols_div_batch_cg_f64), (np.float32, np.complex64, emu.emu_cols_div_batch_cg_f32)):
        prod = np.zeros((ne, 2, my, kp), dtype=cdt)
        prod[..., :W] = rng.standard_normal((ne, 2, my, W)) + 1j * rng.standard_normal((ne, 2, my, W))
        Fn = np.zeros((ne, ny_full, kp), dtype=dt)
        Fn[:, band, :wv] = rng.uniform(0.5, 1.5, (ne, band.size, wv))
        out = np.full((ne, ny_full, kp), 3.0 + 0j, dtype=cdt)
        lxd_t, lyd_t = lxd.astype(dt), lyd.astype(dt)
        assert fn(ny_full, my, nx, _p(prod), _p(Fn), _p(lxd_t), _p(lyd_t), _p(out), ne, w, rb) == 0
        one = emu.emu_cols_div_cg_f64 if dt == np.float64 else emu.emu_cols_div_cg_f32
        for e_ in range(ne):
            ref = np.full((ny_full, kp), 3.0 + 0j, dtype=cdt)
            A = np.ascontiguousarray(prod[e_, 0]); B = np.ascontiguousarray(prod[e_, 1]); F = np.ascontiguousarray(Fn[e_])
            assert one(ny_full, my, nx, _p(A), _p(B), _p(F), _p(lxd_t), _p(lyd_t), _p(ref), w, rb) == 0
            assert np.array_equal(out[e_], ref)


def test_one_wave_per_row_r2c(emu):
    """row_r2c_w64_body (fft_r2c_w64.hpp): 8192-point real rows, one 64-lane wave per row, two radix-64 stages around
    one LDS transpose, pruned second stage -> the first `width` columns of numpy.fft.rfft."""
    ny, nx, width = 5, 8192, 380
    rng = np.random.default_rng(64)
    x = rng.standard_normal((ny, nx)).astype(np.float32)
    kp = emu.emu_kpitch(nx)
    out = np.full((ny, kp), 3.0 + 0j, dtype=np.complex64)
    assert emu.emu_r2c_rows_w64_f32(ny, nx, _p(x), _p(out), ctypes.c_double(0.5), width, 2) == 0
    ref = 0.5 * np.fft.rfft(x.astype(np.float64), axis=1)
    assert np.abs(out[:, :width] - ref[:, :width]).max() < 2e-6 * np.abs(ref).max()
    assert np.all(out[:, width:] == 3.0)
    out2 = np.full((ny, kp), 3.0 + 0j, dtype=np.complex64)
    assert emu.emu_r2c_rows_w64_f32(ny, nx, _p(x), _p(out2), ctypes.c_double(1.0), 512, 3) == 0
    assert np.abs(out2[:, :512] - 2 * ref[:, :512]).max() < 2e-6 * 2 * np.abs(ref).max()


def test_two_waves_per_row_r2c(emu):
    """row_r2c_w64x2_body: 16384-point real rows, even / odd packed samples in two waves, radix-2 combine + untangle."""
    ny, nx, width = 3, 16384, 760
    rng = np.random.default_rng(65)
    x = rng.standard_normal((ny, nx)).astype(np.float32)
    kp = emu.emu_kpitch(nx)
    out = np.full((ny, kp), 3.0 + 0j, dtype=np.complex64)
    assert emu.emu_r2c_rows_w64x2_f32(ny, nx, _p(x), _p(out), ctypes.c_double(1.0), width, 2) == 0
    ref = np.fft.rfft(x.astype(np.float64), axis=1)
    assert np.abs(out[:, :width] - ref[:, :width]).max() < 2e-6 * np.abs(ref).max()
    assert np.all(out[:, width:] == 3.0)


@pytest.mark.parametrize("ny,my,w,rb", [(4096, 1024, 20, 190), (8192, 2048, 33, 300), (4096, 1024, 0, 100)])
def test_fused_forward_pass2_legs_on_the_column_grid(emu, ny, my, w, rb):
    """col_fwdlegs_cg_body (forward pass 2 + leg filter + 16-point inverse pass 1, one leg per workgroup) followed by the
    16 x My/16 inverse pass 2 equals the three-launch column-grid path AND the NumPy inverse transform of the leg band."""
    nx = 64
    rng = np.random.default_rng(ny + w)
    W = nx // 2 + 1
    wv = w if w else W
    kp = emu.emu_kpitch(nx)
    ly = 2 * np.pi * np.fft.fftfreq(ny) * 100
    lx = 2 * np.pi * np.fft.fftfreq(nx) * 100
    lyd, lxd = ly.copy(), lx.copy()
    lyd[ny // 2] = 0
    lxd[nx // 2] = 0
    x = rng.standard_normal((ny, nx))
    band = np.r_[0:rb, ny - rb + 1:ny]
    FG = np.zeros((ny, kp)); FH = np.zeros((ny, kp))
    FG[band, :wv] = rng.uniform(0.5, 1.5, (band.size, wv))
    FH[band, :wv] = rng.uniform(0.5, 1.5, (band.size, wv))
    outs = {}
    for fused in (1, 0):
        o = [_hc(emu, my, nx, fill=3.0) for _ in range(3)]
        assert emu.emu_map_legs_cols_cg_f64(ny, my, nx, _p(x), _p(FG), _p(FH), _p(lxd), _p(lyd), _p(o[0]), _p(o[1]), _p(o[2]), w, rb, fused) == 0
        outs[fused] = o
    rows = np.r_[0:my // 2, ny - my // 2:ny]
    kT = np.fft.rfft2(x)[rows]
    lx2, ly2 = lxd[None, :W], lyd[rows][:, None]
    refs = [_col_ifft(1j * lx2 * FG[rows][:, :W] * kT, my), _col_ifft(1j * ly2 * FG[rows][:, :W] * kT, my), _col_ifft(FH[rows][:, :W] * kT, my)]
    for a, b, want in zip(outs[1], outs[0], refs):
        assert np.abs(a[:, :wv] - want[:, :wv]).max() < 1e-11 * np.abs(want).max()
        assert np.abs(a[:, :wv] - b[:, :wv]).max() < 1e-11 * np.abs(want).max()
        if w:
            assert np.all(a[:, wv:W] == 3.0)


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("ny_full,my,w,rb,nx", [(4096, 1024, 20, 150, 64), (8192, 2048, 0, 300, 64), (8192, 2048, 200, 300, 512), (8192, 4096, 70, 700, 256)])
def test_single_pass_forward_columns_and_divergence(emu, ny_full, my, w, rb, nx, prec):
    """col_div_body with a whole 1024- / 2048- / 4096-point column in the tile (f32: 16 / 8 / 4 columns, 1024 threads; f64: 8 / 4 / 2
    columns, 512 threads -- 128 KB of LDS either way; 4096 rows: the wide band's column grid at 8192^2): forward column transform + divergence in ONE pass, with the
    column-grid row mapping of Fn, ly and the output; the 512-column case has more than 16 tiles, i.e. exercises the
    XCD-pairing tile order of the 64-byte-segment tiles."""
    rdt, cdt = (np.float32, np.complex64) if prec == "f32" else (np.float64, np.complex128)
    fn = emu.emu_cols_div_cg_f32 if prec == "f32" else emu.emu_cols_div_cg_f64
    rng = np.random.default_rng(my + w)
    W = nx // 2 + 1
    wv = w if w else W
    kp = emu.emu_kpitch(nx)
    ly = (2 * np.pi * np.fft.fftfreq(ny_full) * 100).astype(rdt)
    lx = (2 * np.pi * np.fft.fftfreq(nx) * 100).astype(rdt)
    lyd, lxd = ly.copy(), lx.copy()
    lyd[ny_full // 2] = 0
    lxd[nx // 2] = 0
    rows = np.r_[0:my // 2, ny_full - my // 2:ny_full]
    band = np.r_[0:rb, ny_full - rb + 1:ny_full]
    Fn = np.zeros((ny_full, kp), dtype=rdt)
    Fn[band, :wv] = rng.uniform(0.5, 1.5, (band.size, wv)).astype(rdt)
    A = np.zeros((my, kp), dtype=cdt); B = np.zeros((my, kp), dtype=cdt)
    A[:, :W] = (rng.standard_normal((my, W)) + 1j * rng.standard_normal((my, W))).astype(cdt)
    B[:, :W] = (rng.standard_normal((my, W)) + 1j * rng.standard_normal((my, W))).astype(cdt)
    out = np.full((ny_full, kp), 3.0 + 0j, dtype=cdt)
    assert fn(ny_full, my, nx, _p(A), _p(B), _p(Fn), _p(lxd), _p(lyd), _p(out), w, rb) == 0
    lx2, ly2 = lxd[None, :W].astype(np.float64), lyd[rows][:, None].astype(np.float64)
    want = Fn[rows][:, :W] * (1j * lx2 * np.fft.fft(A[:, :W].astype(np.complex128), axis=0) + 1j * ly2 * np.fft.fft(B[:, :W].astype(np.complex128), axis=0))
    cb = np.r_[0:rb, my - rb + 1:my]
    assert np.abs(out[rows[cb]][:, :wv] - want[cb][:, :wv]).max() < (2e-5 if prec == "f32" else 1e-12) * np.abs(want).max()
    untouched = np.setdiff1d(np.arange(ny_full), rows[cb])
    assert np.all(out[untouched] == 3.0)
    if w:
        assert np.all(out[:, wv:W] == 3.0)


# ---------------------------------------------------------------------------------------------------------------
# R-SPLIT from-map path: row R2C with the first radix-4 butterfly of the column transform (general pass and the
# one-wave-per-row kernel), the single-pass column stage (fft_fband.hpp) and the row stage on its R-LAYOUT planes
# ---------------------------------------------------------------------------------------------------------------
def _rsplit_reference(x, my, w):
    """Y[k1][g][k] = W_ny^(g k1) sum_n rfft(x[g + my n])[k] W_R^(n k1), k < w, R = ny / my"""
    ny = x.shape[0]
    R = ny // my
    X = np.fft.rfft(x, axis=1)[:, :w]
    g = np.arange(my)
    out = np.zeros((R, my, w), dtype=np.complex128)
    for k1 in range(R):
        acc = sum(X[n * my:(n + 1) * my] * np.exp(-2j * np.pi * n * k1 / float(R)) for n in range(R))
        out[k1] = acc * np.exp(-2j * np.pi * g * k1 / ny)[:, None]
    return out


@pytest.mark.parametrize("prec,pf", [("f64", 0), ("f32", 0), ("f64", 1)])
def test_rsplit_row_pass(emu, prec, pf):
    emu.emu_set_rsplit_pf(pf)                  # 1: persistent workgroups, next row's taps loaded before the current row's later stages
    ny, my, nx, w = 4096, 1024, 2048, 37
    rdt, cdt, tol = (np.float64, np.complex128, 1e-12) if prec == "f64" else (np.float32, np.complex64, 3e-6)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((ny, nx)).astype(rdt)
    pitch = 64
    Y = np.full((4, my, pitch), 7.0 + 0j, dtype=cdt)
    fn = emu.emu_rsplit_rows_f64 if prec == "f64" else emu.emu_rsplit_rows_f32
    assert fn(ny, my, nx, _p(x), _p(Y), ctypes.c_long(pitch), w) == 0
    ref = _rsplit_reference(x.astype(np.float64), my, w)
    assert np.abs(Y[:, :, :w] - ref).max() < tol * np.abs(ref).max()
    emu.emu_set_rsplit_pf(0)
    assert np.all(Y[:, :, w:] == 7.0)                                     # nothing beyond the kept columns is written
    # ... and the column transform of plane k1 over g gives the full-resolution modes k1 + 4 k2
    full = np.fft.fft(np.fft.rfft(x.astype(np.float64), axis=1)[:, :w], axis=0)
    for k1 in range(4):
        assert np.abs(np.fft.fft(ref[k1], axis=0) - full[k1::4]).max() < 1e-9 * np.abs(full).max()


@pytest.mark.parametrize("nx,prec,w,pf", [(8192, "f64", 380, 1), (8192, "f64", 512, 0), (8192, "f32", 380, 1), (8192, "f64", 1, 1),
                                          (4096, "f64", 190, 0), (4096, "f32", 256, 1), (4096, "f64", 256, 1), (4096, "f32", 3, 0),
                                          (16384, "f64", 380, 1), (16384, "f64", 512, 0), (16384, "f32", 380, 1), (16384, "f32", 2, 0),
                                          (8192, "f64", 1138, 0), (8192, "f32", 1138, 1), (8192, "f64", 1280, 1), (8192, "f32", 513, 0)])
def test_rsplit_row_pass_one_crosswave_exchange(emu, nx, prec, w, pf):
    """row_r2c_rs_body: L = 16 x S with the sub-transforms (S = 256 = 16 x 16 points for 8192-point rows, 128 = 16 x 8 for
    4096-point rows, 512 = 16 x 32 for 16384-point rows: lane pairs share a 32-point butterfly) inside one wave's part of the
    buffer, pruned last stage, persistent workgroups (3 walk 8 groups) with and without the prefetch order; the column butterfly
    on top is R = 4, for 16384-point rows R = 8; more than 512 kept columns of 8192-point rows: the wide band (five kept bins per side
    of the last stage, five columns per thread) with R = 2"""
    R = 8 if nx == 16384 else (2 if w > 512 else 4)
    ny = 8 * R
    my = ny // R
    rdt, cdt, tol = (np.float64, np.complex128, 1e-12) if prec == "f64" else (np.float32, np.complex64, 3e-6)
    rng = np.random.default_rng(11)
    x = rng.standard_normal((ny, nx)).astype(rdt)
    pitch = 520 if w <= 512 else 1296
    Y = np.full((R, my, pitch), 7.0 + 0j, dtype=cdt)
    fn = emu.emu_rsplit_rows_rs4096_f64 if prec == "f64" else emu.emu_rsplit_rows_rs4096_f32
    assert fn(ny, nx, _p(x), _p(Y), ctypes.c_long(pitch), w, 3, pf) == 0
    ref = _rsplit_reference(x.astype(np.float64), my, w)
    assert np.abs(Y[:, :, :w] - ref).max() < tol * np.abs(ref).max()
    assert np.all(Y[:, :, w:] == 7.0)


def _fband_reference(Y, my, FG, FH, lxd, lyd, w, ny):
    """leg planes in the R-LAYOUT (row y_lo R + k1) from the row pass's planes Y[k1][g][k], R = ny / my"""
    R = ny // my
    mq = my // R
    full = np.zeros((ny, w), dtype=np.complex128)
    for k1 in range(R):
        full[k1::R] = np.fft.fft(Y[k1][:, :w], axis=0)
    legs = [full * FH[:, :w], 1j * lxd[None, :w] * FG[:, :w] * full, 1j * lyd[:, None] * FG[:, :w] * full]    # H, Gx, Gy
    outs, fields = [], []
    for leg in legs:
        coarse = np.concatenate([leg[:my // 2], leg[ny - my // 2:]])            # the My-row spectrum (band-limited legs)
        fields.append(np.fft.ifft(coarse, axis=0) * my)
        plane = np.zeros((my, w), dtype=np.complex128)
        ylo = np.arange(mq)
        for k1 in range(R):
            B = np.fft.ifft(coarse[k1::R], axis=0) * mq * np.exp(2j * np.pi * k1 * ylo / my)[:, None]
            plane[k1::R] = B                                                     # row y_lo * R + k1
        outs.append(plane)
    return outs, fields


@pytest.mark.parametrize("packed", [0, 1], ids=["planes", "packed"])
@pytest.mark.parametrize("prec,nmaps,ny,my", [("f64", 1, 4096, 1024), ("f32", 2, 4096, 1024), ("f64", 1, 16384, 2048), ("f32", 1, 16384, 2048),
                                              ("f64", 1, 8192, 4096), ("f32", 2, 8192, 4096)])
def test_rsplit_single_pass_column_stage(emu, prec, nmaps, ny, my, packed):
    """col_fband_body (R = 4, 8, 2) against NumPy; packed: the filters through the per-binding table of col_fband_pack_body (what the
    one-call entries run) instead of the filter planes"""
    nx, w, rb = 2048, 21, (150 if my < 4096 else 1139)
    R = ny // my
    rdt, cdt, tol = (np.float64, np.complex128, 1e-12) if prec == "f64" else (np.float32, np.complex64, 4e-6)
    rng = np.random.default_rng(9)
    kp = emu.emu_kpitch(nx)
    pitch, opitch = 32, 32
    ly = (2 * np.pi * np.fft.fftfreq(ny) * 100)
    lx = (2 * np.pi * np.fft.fftfreq(nx) * 100)
    lyd, lxd = ly.copy(), lx.copy()
    lyd[ny // 2] = 0
    lxd[nx // 2] = 0
    band = np.r_[0:rb, ny - rb + 1:ny]
    FG = np.zeros((ny, kp)); FH = np.zeros((ny, kp))
    FG[band, :w] = rng.uniform(0.5, 1.5, (band.size, w))
    FH[band, :w] = rng.uniform(0.5, 1.5, (band.size, w))
    Y = np.zeros((nmaps, R, my, pitch), dtype=cdt)
    Y[..., :w] = (rng.standard_normal((nmaps, R, my, w)) + 1j * rng.standard_normal((nmaps, R, my, w))).astype(cdt)
    outs = [np.full((nmaps, my, opitch), 5.0 + 0j, dtype=cdt) for _ in range(3)]           # gx, gy, h
    fn = emu.emu_rsplit_legs_f64 if prec == "f64" else emu.emu_rsplit_legs_f32
    args = [a.astype(rdt) for a in (FG, FH, lxd, lyd)]
    emu.emu_set_fband_packed(packed)
    assert fn(ny, my, nx, _p(Y), ctypes.c_long(pitch), _p(args[0]), _p(args[1]), _p(args[2]), _p(args[3]), _p(outs[0]), _p(outs[1]), _p(outs[2]),
              ctypes.c_long(opitch), w, rb, nmaps, ctypes.c_long(R * my * pitch), ctypes.c_long(my * opitch)) == 0
    emu.emu_set_fband_packed(0)
    for m in range(nmaps):
        (rh, rgx, rgy), fields = _fband_reference(Y[m].astype(np.complex128), my, FG, FH, lxd, lyd, w, ny)
        for got, want in ((outs[2][m], rh), (outs[0][m], rgx), (outs[1][m], rgy)):
            assert np.abs(got[:, :w] - want).max() < tol * np.abs(want).max()
            assert np.all(got[:, w:] == 5.0)
        # the R-layout really encodes the field: x[y_lo + Mq y_hi] = sum_k1 W_R^(-k1 y_hi) B[k1][y_lo]
        mq = my // R
        B = rh.reshape(mq, R, w)
        x = np.stack([sum(B[:, k1] * np.exp(2j * np.pi * k1 * yh / float(R)) for k1 in range(R)) for yh in range(R)]).reshape(my, w)
        assert np.abs(x - fields[0]).max() < 1e-9 * np.abs(fields[0]).max()


@pytest.mark.parametrize("prec,R,M,win,wout", [("f64", 4, 1024, 20, 30), ("f32", 4, 1024, 20, 30), ("f64", 8, 2048, 20, 30), ("f32", 8, 2048, 20, 30),
                                               ("f64", 8, 2048, 380, 664), ("f64", 4, 2048, 380, 664), ("f64", 4, 1536, 380, 664),
                                               ("f32", 4, 1536, 380, 664), ("f64", 8, 1536, 380, 664), ("f64", 4, 4096, 1139, 664),
                                               ("f64", 2, 4096, 1139, 664), ("f32", 2, 4096, 1139, 664), ("f64", 2, 4096, 600, 664)])
def test_row_stage_on_r_layout_planes(emu, body, prec, R, M, win, wout):
    """row_qe_pair_body<.., LAY = 2 / 3>: the same products from leg planes in the R-LAYOUT (R = 4, R = 8) as from natural-order planes
    (380 / 664 columns: the headline's band limits -- four live taps per side of the first inverse stage)"""
    if body == 16 and (M == 1536 or R == 2):
        pytest.skip("the 3 x 512-point grid and the R = 2 layout exist in the 8-point body only")
    my, nx = 64, (8192 if M == 4096 else 4096)
    rdt, cdt, tol = (np.float64, np.complex128, 1e-12) if prec == "f64" else (np.float32, np.complex64, 2e-5)
    rng = np.random.default_rng(12)
    kp = emu.emu_kpitch(nx)
    nat = []
    for _ in range(3):
        a = np.zeros((my, kp), dtype=np.complex128)
        a[:, :win] = rng.standard_normal((my, win)) + 1j * rng.standard_normal((my, win))
        a[:, 0] = a[:, 0].real                       # (a real row's transform is real at k = 0)
        nat.append(a)
    mq = my // R

    def to_r(a):                                     # B[k1][y_lo] = (1/R) sum_yh x[y_lo + Mq yh] W_R^(k1 yh), stored at row y_lo R + k1
        x = a.reshape(R, mq, kp)
        out = np.zeros_like(a)
        for k1 in range(R):
            out[k1::R] = sum(x[yh] * np.exp(-2j * np.pi * k1 * yh / float(R)) for yh in range(R)) / float(R)
        return out
    fn = emu.emu_qe_rows_rlayout_f64 if prec == "f64" else emu.emu_qe_rows_rlayout_f32
    res = []
    for lr, planes in ((0, nat), ({2: 1, 4: 2, 8: 3}[R], [to_r(a) for a in nat])):
        pl = [np.ascontiguousarray(a.astype(cdt)) for a in planes]
        px = np.full((my, kp), 3.0 + 0j, dtype=cdt); py = np.full((my, kp), 3.0 + 0j, dtype=cdt)
        assert fn(my, nx, _p(pl[0]), _p(pl[1]), _p(pl[2]), _p(px), _p(py), ctypes.c_double(1.0 / nx ** 2), win, wout, M, lr) == 0
        res.append((px, py))
    for a, b in zip(res[0], res[1]):
        assert np.abs(a[:, :wout] - b[:, :wout]).max() < tol * np.abs(a[:, :wout]).max()
        assert np.all(b[:, wout:] == 3.0)


@pytest.mark.parametrize("ny,nx,wc,prec", [(8, 4096, 190, "f64"), (16, 1024, 513, "f64"), (8, 8192, 380, "f32"), (4, 16384, 300, "f64")])
def test_fused_windowed_row_pass(emu, ny, nx, wc, prec):
    """ROW_WIN: half-complex rows -> C2R -> x real-space window -> R2C (kept columns only) in one pass, the real rows in LDS only
    (oa_mc_run_windowed; maps.py:1350-1361 multiplies every map by its taper before the transform)"""
    rdt, cdt, tol = (np.float64, np.complex128, 1e-12) if prec == "f64" else (np.float32, np.complex64, 4e-6)
    rng = np.random.default_rng(23)
    kp = emu.emu_kpitch(nx)
    x = rng.standard_normal((ny, nx))
    hc = np.zeros((ny, kp), dtype=cdt)
    hc[:, :nx // 2 + 1] = np.fft.rfft(x, axis=1)
    w = (0.5 + rng.uniform(size=(ny, nx))).astype(rdt)
    opitch = 640
    out = np.full((ny, opitch), 7.0 + 0j, dtype=cdt)
    fn = emu.emu_rows_win_f64 if prec == "f64" else emu.emu_rows_win_f32
    assert fn(ny, nx, _p(hc), _p(w), _p(out), ctypes.c_long(opitch), ctypes.c_double(1.0 / nx), wc) == 0
    ref = np.fft.rfft(x * w.astype(np.float64), axis=1)[:, :wc]
    assert np.abs(out[:, :wc] - ref).max() < tol * np.abs(ref).max()
    assert np.all(out[:, wc:] == 7.0)


@pytest.mark.parametrize("ny,nx,order,separable", [(64, 128, 5, 1), (256, 64, 3, 1), (64, 128, 5, 0), (32, 256, 6, 1), (64, 128, 5, 2), (32, 64, 1, 2)])
def test_batched_derivative_inverse_transforms(emu, ny, nx, order, separable):
    """col_deriv_body + batched row C2R (oa_lens_maps): every Fourier-space derivative (i lx)^a (i ly)^b k, a + b < order, of a
    transform inverse-transformed with the factor applied at the load of the inverse column pass; plane idx(a, b) =
    n (n + 1) / 2 - 1 + b, n = a + b (the order lens_taylor_kernel reads).  separable (what oa_lens_maps runs): one column transform per
    y-derivative order b, the x-derivative orders as ONE row launch with (i lx)^a at the load of the C2R"""
    rng = np.random.default_rng(31)
    kp = emu.emu_kpitch(nx)
    x = rng.standard_normal((ny, nx))
    k0 = np.zeros((ny, kp), dtype=np.complex128)
    k0[:, :nx // 2 + 1] = np.fft.rfft2(x)
    ly = 2 * np.pi * np.fft.fftfreq(ny) * 3.0
    lx = 2 * np.pi * np.fft.fftfreq(nx) * 5.0
    lyd, lxd = ly.copy(), lx.copy()
    lyd[ny // 2] = 0
    lxd[nx // 2] = 0
    nd = order * (order + 1) // 2 - 1
    d00 = 1 if separable == 2 else 0          # oa_lens_maps_hc: the field itself is plane 0 of nd + 1, out of the b = 0 row launch
    out = np.zeros((nd + d00, ny, nx))
    assert emu.emu_lens_derivs_f64(ny, nx, _p(k0), _p(lxd), _p(lyd), _p(out), nd, separable) == 0
    kf = np.fft.rfft2(x)
    if d00:
        assert np.abs(out[0] - x).max() < 1e-12 * np.abs(x).max()
        out = out[1:]
    for n in range(1, order):
        for b in range(n + 1):
            a = n - b
            ref = np.fft.irfft2(kf * (1j * lxd[None, :nx // 2 + 1]) ** a * (1j * lyd[:, None]) ** b, s=(ny, nx))
            got = out[n * (n + 1) // 2 - 1 + b]
            assert np.abs(got - ref).max() < 1e-11 * max(np.abs(ref).max(), 1e-30), (a, b)


def test_c2r_drops_the_non_hermitian_part_of_the_self_conjugate_columns(emu):
    """C2R of a half-complex plane whose kx = 0 and kx = nx/2 columns are NOT Hermitian in y (e.g. Q, U = R^-1 (E, B): the
    rotation's sine is odd on the Nyquist column): the result is the real part of the inverse transform of the Hermitian-completed
    plane, column by column -- numpy.fft.irfft2's and the reference's `ifft(...).real` semantics (maps.py:1585) -- and in
    particular the Nyquist column's antisymmetric part does not leak into kx = 0."""
    ny, nx = 32, 64
    rng = np.random.default_rng(41)
    kp = emu.emu_kpitch(nx)
    hc = np.zeros((ny, kp), dtype=np.complex128)
    hc[:, :nx // 2 + 1] = rng.standard_normal((ny, nx // 2 + 1)) + 1j * rng.standard_normal((ny, nx // 2 + 1))
    out = np.zeros((ny, nx))
    assert emu.emu_c2r_f64(ny, nx, _p(hc), _p(out), ctypes.c_double(1.0 / (ny * nx))) == 0
    ref = np.fft.irfft2(hc[:, :nx // 2 + 1], s=(ny, nx))
    assert np.abs(out - ref).max() < 1e-13 * np.abs(ref).max()
    back = np.fft.rfft2(out)
    herm0 = 0.5 * (hc[:, 0] + np.conj(hc[(-np.arange(ny)) % ny, 0]))
    assert np.abs(back[:, 0] - herm0).max() < 1e-12 * np.abs(herm0).max()      # kx = 0 holds ITS OWN Hermitian part only


@pytest.mark.parametrize("ny,nx", [(12, 20), (30, 50), (60, 36), (40, 600), (150, 24), (16, 1200), (90, 120)])
@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_mixed_radix_passes_match_numpy(emu, ny, nx, prec):
    """fft_mixed.hpp (sides 2^a 3^b 5^c that are not powers of two -- the reference notebooks' 600 / 1200 / 2400-pixel patches): the
    radix-2/3/4/5 Stockham stages, the packed real-row transform with its (un)tangle and the column tiles, against numpy.fft."""
    rdt, cdt, tol = (np.float64, np.complex128, 1e-12) if prec == "f64" else (np.float32, np.complex64, 3e-6)
    fn = emu.emu_mixed_f64 if prec == "f64" else emu.emu_mixed_f32
    rng = np.random.default_rng(ny * 131 + nx)
    x = rng.standard_normal((ny, nx)).astype(rdt)
    k = np.zeros((ny, nx // 2 + 1), dtype=cdt)
    assert fn(ny, nx, 0, _p(x), _p(k)) == 0
    ref = np.fft.rfft2(x.astype(np.float64))
    assert np.abs(k - ref).max() < tol * np.abs(ref).max() * np.log2(ny * nx)
    back = np.zeros((ny, nx), dtype=rdt)
    kin = np.ascontiguousarray(ref.astype(cdt))
    kin[:, 0] += 0.5j * rng.standard_normal(ny).astype(rdt)          # a non-Hermitian k_x = 0 column: its imaginary part is dropped, as ifft(...).real drops it
    assert fn(ny, nx, 1, _p(kin), _p(back)) == 0
    full = np.zeros((ny, nx), dtype=np.complex128)
    full[:, :nx // 2 + 1] = kin
    full[:, nx // 2 + 1:] = np.conj(np.roll(kin[::-1], 1, axis=0)[:, 1:nx // 2][:, ::-1])
    want = np.fft.ifft2(full).real * ny * nx
    # (the packed C2R keeps the Hermitian part of the self-conjugate columns only: compare with the same symmetrisation)
    sym = full.copy()
    for c in (0, nx // 2):
        col = full[:, c]
        sym[:, c] = 0.5 * (col + np.conj(np.roll(col[::-1], 1)))
    want = np.fft.ifft2(sym).real * ny * nx
    assert np.abs(back - want).max() < tol * np.abs(want).max() * np.log2(ny * nx)
    z = (rng.standard_normal((ny, nx)) + 1j * rng.standard_normal((ny, nx))).astype(cdt)
    o = np.zeros_like(z)
    assert fn(ny, nx, 2, _p(z), _p(o)) == 0
    assert np.abs(o - np.fft.fft2(z.astype(np.complex128))).max() < tol * np.abs(o).max() * np.log2(ny * nx)
    assert fn(ny, nx, 3, _p(z), _p(o)) == 0
    assert np.abs(o - np.fft.ifft2(z.astype(np.complex128)) * ny * nx).max() < tol * np.abs(o).max() * np.log2(ny * nx)

