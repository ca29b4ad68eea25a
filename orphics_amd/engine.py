"""Thin device layer: one :class:`Engine` = one C-ABI plan (geometry, dtype,
GPU) + typed wrappers that validate tensor shapes on the host before any kernel
is launched.  torch is used only for device memory and streams."""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import OA_F32, OA_F64, check


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dirty(t):
    mark_dirty(t)
    return t


_RDT = {"f32": torch.float32, "f64": torch.float64}
_CDT = {"f32": torch.complex64, "f64": torch.complex128}
_CODE = {"f32": OA_F32, "f64": OA_F64}


def precision_of(x, default="f64"):
    """dtype follows the data: float32/complex64 -> 'f32', float64/complex128 -> 'f64'."""
    dt = getattr(x, "dtype", None)
    if dt in (np.float32, np.complex64, torch.float32, torch.complex64):
        return "f32"
    if dt in (np.float64, np.complex128, torch.float64, torch.complex128):
        return "f64"
    return default


_BIN_SCRATCH = {}

# ---- estimator-owned output planes ------------------------------------------------------------------------
# The pruned estimator kernels write only the active region of kappa_hat (leading columns x row band); the rest of
# the plane must already be zero.  Re-zeroing ~A bytes per reconstruction would cost more than the divergence
# kernel itself, so planes handed out by ``lensing.Estimator.new_output()`` are tracked here: (weakref, region the
# complement of which is known to be zero, torch version counter at that time).  ANY write through an Engine wrapper
# (``out=`` arguments below call :func:`mark_dirty`) or through an in-place torch op (bumps ``_version``) drops the
# guarantee and the next reconstruction zero-fills again.  Planes that did not come from ``new_output()`` are never
# trusted: their complement is zeroed on every call.
_OWNED = {}


def register_owned(t):
    import weakref
    if len(_OWNED) > 256:
        for k in [k for k, v in _OWNED.items() if v[0]() is None]:
            del _OWNED[k]
    _OWNED[t.data_ptr()] = [weakref.ref(t), None, t._version]
    return t


def mark_dirty(t):
    """Forget what is known about the contents of plane ``t`` (called by every wrapper that writes into ``out=``)."""
    if t is None:
        return
    r = _OWNED.get(t.data_ptr())
    if r is not None:
        r[1] = None


def owned_clean_region(t):
    """Region whose complement is known zero for an estimator-owned plane, else None."""
    r = _OWNED.get(t.data_ptr())
    if r is None or r[0]() is not t or r[2] != t._version:
        return None
    return r[1]


def set_clean_region(t, region):
    r = _OWNED.get(t.data_ptr())
    if r is not None and r[0]() is t:
        r[1] = region
        r[2] = t._version
        return True
    return False


def cuda_device():
    _lib.require_gpu()
    if not torch.cuda.is_available():
        raise _lib.OrphicsAmdError("orphics_amd: torch sees no GPU; device memory cannot be allocated")
    return torch.device("cuda", torch.cuda.current_device())


def dev_digitize(x64, edges64):
    """np.digitize(x, edges, right=True) on device (float64, bit-exact); plan-free."""
    lib = _lib.load()
    if x64.dtype != torch.float64 or edges64.dtype != torch.float64 or not x64.is_cuda or not edges64.is_cuda:
        raise ValueError("digitize operates on float64 CUDA tensors")
    x64 = x64.contiguous()
    ids = torch.empty(x64.shape, dtype=torch.int32, device=x64.device)
    check(lib.oa_digitize(_ptr(x64), x64.numel(), _ptr(edges64.contiguous()), edges64.numel(), _ptr(ids), _stream()))
    return ids


def dev_bin(data, ids, nids, weights=None, aux=None, mode=0, skip_nan=False, herm_pitch=0, herm_nxh=-1):
    """Streaming histogram (oa_bin); returns (sums f64[nids], counts int64[nids] | wsums f64[nids])."""
    lib = _lib.load()
    n = data.numel()
    if not (data.is_cuda and ids.is_cuda):
        raise TypeError("bin: CUDA tensors required")
    if ids.numel() != n or ids.dtype != torch.int32 or not ids.is_contiguous() or not data.is_contiguous():
        raise ValueError("bin: ids must be contiguous int32 with one entry per data element")
    prec = precision_of(data, default=None)
    if prec is None:
        raise ValueError("bin: data must be float32 or float64")
    if weights is not None and (weights.dtype != data.dtype or weights.numel() != n or not weights.is_contiguous()):
        raise ValueError("bin: weights must match data")
    if aux is not None and (aux.dtype != torch.float64 or aux.numel() != nids):
        raise ValueError("bin: aux must be float64[nids]")
    need = int(lib.oa_bin_scratch_bytes(int(nids)))
    if need < 0:
        raise ValueError("bin: bad nids")
    key = (data.device.index, torch.cuda.current_stream().cuda_stream)   # one scratch per stream: streams may overlap
    scr = _BIN_SCRATCH.get(key)
    if scr is None or scr.numel() < need:
        scr = torch.empty(need, dtype=torch.uint8, device=data.device)
        _BIN_SCRATCH[key] = scr
    sums = torch.empty(nids, dtype=torch.float64, device=data.device)
    counts = torch.empty(nids, dtype=torch.int64, device=data.device) if weights is None else None
    wsums = torch.empty(nids, dtype=torch.float64, device=data.device) if weights is not None else None
    check(lib.oa_bin(_CODE[prec], _ptr(data), _ptr(ids), _ptr(weights), _ptr(aux), n, int(nids), int(mode),
                     1 if skip_nan else 0, int(herm_pitch), int(herm_nxh), _ptr(sums), _ptr(counts), _ptr(wsums),
                     _ptr(scr), _stream()))
    return sums, (counts if weights is None else wsums)


def dev_bin_power(k1, k2, norm, ids, nids, herm_pitch=0, herm_nxh=-1, active_cols=0, active_rows=0):
    """Binned Re(conj k1 k2)*norm without materialising the 2-D power (oa_bin_power).
    ``active_cols`` > 0: visit only those leading columns of each row (planes that vanish beyond them);
    ``active_rows`` > 0: and only the rows y < active_rows or y > ny - active_rows;
    the returned counts then cover the visited region only."""
    lib = _lib.load()
    n = k1.numel()
    if not (k1.is_cuda and k2.is_cuda and ids.is_cuda and k1.is_complex() and k2.dtype == k1.dtype):
        raise TypeError("bin_power: complex CUDA tensors of equal dtype required")
    if k2.numel() != n or ids.numel() != n or ids.dtype != torch.int32 or not (ids.is_contiguous() and k1.is_contiguous() and k2.is_contiguous()):
        raise ValueError("bin_power: operands must be contiguous with one int32 id per mode")
    prec = precision_of(k1, default=None)
    need = int(lib.oa_bin_scratch_bytes(int(nids)))
    key = (k1.device.index, torch.cuda.current_stream().cuda_stream)
    scr = _BIN_SCRATCH.get(key)
    if scr is None or scr.numel() < need:
        scr = torch.empty(need, dtype=torch.uint8, device=k1.device)
        _BIN_SCRATCH[key] = scr
    sums = torch.empty(nids, dtype=torch.float64, device=k1.device)
    counts = torch.empty(nids, dtype=torch.int64, device=k1.device)
    check(lib.oa_bin_power(_CODE[prec], _ptr(k1), _ptr(k2), float(norm), _ptr(ids), None, n, int(nids), int(herm_pitch),
                           int(herm_nxh), _ptr(sums), _ptr(counts), None, _ptr(scr), int(active_cols), int(active_rows), _stream()))
    return sums, counts


class Engine(object):
    _cache = {}

    @classmethod
    def get(cls, ny, nx, prec):
        dev = torch.cuda.current_device() if torch.cuda.is_available() else -1
        key = (int(ny), int(nx), prec, dev)
        e = cls._cache.get(key)
        if e is None:
            e = cls(ny, nx, prec)
            cls._cache[key] = e
        return e

    def __init__(self, ny, nx, prec="f32"):
        self.lib = _lib.load()
        _lib.require_gpu()
        if not torch.cuda.is_available():
            raise _lib.OrphicsAmdError("orphics_amd: torch sees no GPU; device memory cannot be allocated")
        self.ny, self.nx, self.prec = int(ny), int(nx), prec
        self.code = _CODE[prec]
        self.rdt, self.cdt = _RDT[prec], _CDT[prec]
        self.device = torch.device("cuda", torch.cuda.current_device())
        h = ctypes.c_void_p()
        check(self.lib.oa_plan_create(self.ny, self.nx, self.code, ctypes.byref(h)))
        self.plan = h
        self.kp = int(self.lib.oa_plan_kpitch(h))
        self.nxh = self.nx // 2
        self.npix = self.ny * self.nx
        # power-of-two sides: LDS FFT kernels + fused estimator kernels; other even sides: chirp-z FFTs only
        self.pow2 = (self.ny & (self.ny - 1)) == 0 and (self.nx & (self.nx - 1)) == 0

        def smooth(n):
            for r in (2, 3, 5):
                while n % r == 0:
                    n //= r
            return n == 1
        # sides 2^a 3^b 5^c that are not powers of two: mixed-radix transforms (csrc/fft_mixed.hpp) and the one-call lensing op;
        # other even sides: chirp-z transforms, modular calls only
        self.mixed = (not self.pow2) and self.ny % 2 == 0 and self.nx % 2 == 0 and smooth(self.ny) and smooth(self.nx)
        self._laxes = None
        self._bin_scratch = None
        self._last_stream = None          # the plan's scratch planes are single-buffered: see _ordered()

    def __del__(self):
        try:
            if getattr(self, "plan", None):
                self.lib.oa_plan_destroy(self.plan)
                self.plan = None
        except Exception:
            pass

    def _ordered(self):
        """One plan = one set of scratch planes.  Calls arriving on a DIFFERENT torch stream than the previous call
        are ordered behind it (``wait_stream``), so two streams sharing an Engine serialise on the scratch instead
        of racing on it; handles meant to overlap use their own plan (``lensing.Estimator.fork``)."""
        cur = torch.cuda.current_stream()
        last = self._last_stream
        if last is not None and last != cur:
            cur.wait_stream(last)
        self._last_stream = cur

    # ---- allocation -------------------------------------------------------
    def real(self, *lead):
        return torch.empty(tuple(lead) + (self.ny, self.nx), dtype=self.rdt, device=self.device)

    def hc(self, *lead):
        return torch.zeros(tuple(lead) + (self.ny, self.kp), dtype=self.cdt, device=self.device)

    def hc_fresh(self):
        """an hc plane for a kernel that writes every valid column (0 .. nx/2): only the row padding beyond is zero-filled
        (15 of nx/2 + 16 columns) instead of the whole plane -- a 134 MB fill per 4096^2 complex128 plane otherwise"""
        t = torch.empty((self.ny, self.kp), dtype=self.cdt, device=self.device)
        t[:, self.nxh + 1:] = 0
        return t

    def hcreal(self, *lead):
        return torch.zeros(tuple(lead) + (self.ny, self.kp), dtype=self.rdt, device=self.device)

    def full(self, *lead):
        return torch.empty(tuple(lead) + (self.ny, self.nx), dtype=self.cdt, device=self.device)

    def to_real(self, a):
        if isinstance(a, np.ndarray):
            a = np.require(a, requirements=["C", "W"]) if a.flags.writeable else np.array(a, order="C")
        t = torch.as_tensor(a)
        return t.to(device=self.device, dtype=self.rdt).contiguous()

    def to_complex(self, a):
        if isinstance(a, np.ndarray):
            a = np.require(a, requirements=["C", "W"]) if a.flags.writeable else np.array(a, order="C")
        t = torch.as_tensor(a)
        return t.to(device=self.device, dtype=self.cdt).contiguous()

    # ---- validation ---------------------------------------------------------
    def _chk(self, t, kind):
        shp = {"real": (self.ny, self.nx), "hc": (self.ny, self.kp), "hcreal": (self.ny, self.kp),
               "full": (self.ny, self.nx)}[kind]
        dt = self.rdt if kind in ("real", "hcreal") else self.cdt
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise TypeError("expected a CUDA tensor for %s plane" % kind)
        if tuple(t.shape) != shp or t.dtype != dt or not t.is_contiguous():
            raise ValueError("%s plane must be contiguous %s of shape %s, got %s %s" % (kind, dt, shp, t.dtype, tuple(t.shape)))
        return t

    # ---- FFTs ------------------------------------------------------------------
    def rfft(self, x, scale=1.0, out=None, width=0, rband=0):
        """real (ny,nx) -> hc; unnormalised forward (maps.py:1636).  ``width`` > 0: only the first ``width``
        columns of ``out`` are produced (callers that filter with a band-limited mask; see ACTIVE COLUMNS in
        include/orphics_amd.h) -- the rest of ``out`` is left untouched; ``rband`` > 0: only the rows
        y < rband or y > ny - rband of those columns hold the transform, the other rows are undefined."""
        self._ordered()
        self._chk(x, "real")
        if out is None:
            out = self.hc_fresh() if (width <= 0 and rband <= 0 and self.pow2) else self.hc()
        else:
            _dirty(self._chk(out, "hc"))
        check(self.lib.oa_fft_r2c(self.plan, _ptr(x), _ptr(out), float(scale), int(width), int(rband), _stream()))
        return out

    def irfft(self, k, scale=None, out=None, width=0, window=None):
        """hc -> real; default scale 1/Npix (pixell fft.ifft normalize=True, maps.py:1633).
        ``width`` > 0 asserts that columns >= width of ``k`` are zero (they are not read).
        ``window``: real (ny, nx) device plane multiplied into the result inside the last pass (``oa_fft_c2r_windowed``)."""
        self._ordered()
        self._chk(k, "hc")
        out = self.real() if out is None else _dirty(self._chk(out, "real"))
        if scale is None:
            scale = 1.0 / self.npix
        if window is not None:
            self._chk(window, "real")
            check(self.lib.oa_fft_c2r_windowed(self.plan, _ptr(k), _ptr(out), float(scale), _ptr(window), _stream()))
            return out
        check(self.lib.oa_fft_c2r(self.plan, _ptr(k), _ptr(out), float(scale), int(width), _stream()))
        return out

    def cfft(self, z, inverse=False, scale=1.0, out=None):
        self._ordered()
        self._chk(z, "full")
        out = self.full() if out is None else _dirty(self._chk(out, "full"))
        check(self.lib.oa_fft_c2c(self.plan, _ptr(z), _ptr(out), 1 if inverse else 0, float(scale), _stream()))
        return out

    def fft_cols(self, k, inverse=False, scale=1.0, out=None, width=0):
        """Column transforms only (hc -> hc, out != in); ``width`` > 0: first ``width`` columns only."""
        self._ordered()
        self._chk(k, "hc")
        out = self.hc() if out is None else _dirty(self._chk(out, "hc"))
        check(self.lib.oa_fft_cols(self.plan, _ptr(k), _ptr(out), 1 if inverse else 0, float(scale), int(width), _stream()))
        return out

    def qe_rows(self, gx, gy, h, px, py, scale=None, accumulate=False, win=0, wout=0, mrow=0):
        """Fused row stage: P = R2C(C2R(G) * C2R(H)) for G in (gx, gy).  ``win``: leg columns >= win are zero
        (not read); ``wout``: only product columns < wout are written; ``mrow``: row-transform grid (0 = nx,
        -1 = smallest alias-free power of two >= 2 win + wout; include/orphics_amd.h, ROW GRID)."""
        for t in (gx, gy, h, px, py):
            self._chk(t, "hc")
        _dirty(px); _dirty(py)
        if scale is None:
            scale = 1.0 / float(self.npix) ** 2
        check(self.lib.oa_qe_rows(self.plan, _ptr(gx), _ptr(gy), _ptr(h), _ptr(px), _ptr(py), float(scale),
                                  1 if accumulate else 0, int(win), int(wout), int(mrow), _stream()))
        return px, py

    def qe_legs_cols(self, kX, kY, FG, FH, out, width=0, rband=0):
        """Fused leg filters + inverse column transforms (3 planes out, ready for qe_rows); ``width`` > 0:
        the filters vanish for columns >= width, which are neither read nor produced."""
        self._ordered()
        self._chk(kX, "hc"); self._chk(kY, "hc"); self._chk(FG, "hcreal"); self._chk(FH, "hcreal")
        gx, gy, h = out
        for t in out:
            _dirty(self._chk(t, "hc"))
        check(self.lib.oa_qe_legs_cols(self.plan, _ptr(kX), _ptr(kY), _ptr(FG), _ptr(FH), _ptr(gx), _ptr(gy), _ptr(h), int(width), int(rband), _stream()))
        return out

    def qe_map_legs_cols(self, tmap, FG, FH, out, width=0, rband=0):
        """Real map -> the three column-transformed leg planes (both legs from this map): row R2C, forward column
        pass 1, then ONE kernel for forward pass 2 + leg filters + inverse pass 1, then the inverse pass 2."""
        self._ordered()
        self._chk(tmap, "real"); self._chk(FG, "hcreal"); self._chk(FH, "hcreal")
        gx, gy, h = out
        for t in out:
            _dirty(self._chk(t, "hc"))
        check(self.lib.oa_qe_map_legs_cols(self.plan, _ptr(tmap), _ptr(FG), _ptr(FH), _ptr(gx), _ptr(gy), _ptr(h),
                                           int(width), int(rband), _stream()))
        return out

    def qe_cols_div(self, px, py, Fnorm, out=None, accumulate=False, width=0, rband=0):
        """Fused forward column transforms + divergence * normalisation; ``width`` > 0: only the first
        ``width`` columns of ``out`` are produced (Fnorm vanishes beyond them)."""
        self._ordered()
        self._chk(px, "hc"); self._chk(py, "hc"); self._chk(Fnorm, "hcreal")
        out = self.hc() if out is None else _dirty(self._chk(out, "hc"))
        check(self.lib.oa_qe_cols_div(self.plan, _ptr(px), _ptr(py), _ptr(Fnorm), _ptr(out), 1 if accumulate else 0, int(width), int(rband), _stream()))
        return out

    def fft_pass(self, pass_id, src, dst, width=0):
        """Launch one constituent FFT pass (per-kernel timing in bench.py)."""
        check(self.lib.oa_fft_pass(self.plan, int(pass_id), _ptr(src), _ptr(dst), int(width), _stream()))

    # ---- layouts ----------------------------------------------------------------
    def hc_to_full(self, k):
        self._chk(k, "hc")
        out = self.full()
        check(self.lib.oa_hc_to_full(self.plan, _ptr(k), _ptr(out), _stream()))
        return out

    def full_to_hc(self, z):
        self._chk(z, "full")
        out = self.hc()
        check(self.lib.oa_full_to_hc(self.plan, _ptr(z), _ptr(out), _stream()))
        return out

    def hcreal_to_full(self, f):
        self._chk(f, "hcreal")
        out = self.real()
        check(self.lib.oa_hcreal_to_full(self.plan, _ptr(f), _ptr(out), _stream()))
        return out

    def fullreal_to_hc(self, f):
        self._chk(f, "real")
        out = self.hcreal()
        check(self.lib.oa_fullreal_to_hc(self.plan, _ptr(f), _ptr(out), _stream()))
        return out

    # ---- flat elementwise ---------------------------------------------------------
    def _same(self, *ts):
        n = ts[0].numel()
        for t in ts:
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous() and t.numel() == n):
                raise ValueError("elementwise operands must be contiguous CUDA tensors of equal size")
        return n

    def f2power(self, k1, k2, norm, out=None):
        n = self._same(k1, k2)
        if k1.dtype != self.cdt or k2.dtype != self.cdt:
            raise ValueError("f2power: complex dtype mismatch")
        out = torch.empty(k1.shape, dtype=self.rdt, device=self.device) if out is None else _dirty(out)
        check(self.lib.oa_f2power(self.code, _ptr(k1), _ptr(k2), _ptr(out), float(norm), n, _stream()))
        return out

    def cmul_real(self, k, f, out=None):
        n = self._same(k, f)
        if k.dtype != self.cdt or f.dtype != self.rdt:
            raise ValueError("cmul_real: dtype mismatch")
        out = torch.empty_like(k) if out is None else _dirty(out)
        check(self.lib.oa_cmul_real(self.code, _ptr(k), _ptr(f), _ptr(out), n, _stream()))
        return out

    def cmul(self, k, f, out=None):
        """complex * complex, elementwise (k-space filters with a phase)"""
        n = self._same(k, f)
        if k.dtype != self.cdt or f.dtype != self.cdt:
            raise ValueError("cmul: dtype mismatch")
        out = torch.empty_like(k) if out is None else _dirty(out)
        check(self.lib.oa_cmul(self.code, _ptr(k), _ptr(f), _ptr(out), n, _stream()))
        return out

    def mul_real(self, a, b, out=None):
        n = self._same(a, b)
        if a.dtype != self.rdt or b.dtype != self.rdt:
            raise ValueError("mul_real: dtype mismatch")
        out = torch.empty_like(a) if out is None else _dirty(out)
        check(self.lib.oa_mul_real(self.code, _ptr(a), _ptr(b), _ptr(out), n, _stream()))
        return out

    def axpby(self, a, b, alpha, beta, out=None):
        n = self._same(a, b)
        if a.dtype != self.rdt or b.dtype != self.rdt:
            raise ValueError("axpby: dtype mismatch")
        out = torch.empty_like(a) if out is None else _dirty(out)
        check(self.lib.oa_axpby_real(self.code, _ptr(a), _ptr(b), _ptr(out), float(alpha), float(beta), n, _stream()))
        return out

    def rot2(self, c, s, i1, i2):
        n = self._same(c, s, i1, i2)
        if c.dtype != self.rdt or i1.dtype != self.cdt or i2.dtype != self.cdt or s.dtype != self.rdt:
            raise ValueError("rot2: dtype mismatch")
        o1, o2 = torch.empty_like(i1), torch.empty_like(i2)
        check(self.lib.oa_rot2(self.code, _ptr(c), _ptr(s), _ptr(i1), _ptr(i2), _ptr(o1), _ptr(o2), n, _stream()))
        return o1, o2

    # ---- QE legs --------------------------------------------------------------------
    # oa_plan_set_option (include/orphics_amd.h): equivalent launch sequences of the one-call entries of THIS plan
    OPTIONS = {"mc_batch": 1, "mv_batch": 2, "mv_rowbatch": 3, "mv_chain": 4, "div_bin": 5, "win_fused": 6}

    def set_option(self, name, value):
        if name not in self.OPTIONS:
            raise ValueError("unknown plan option %r (one of %s)" % (name, sorted(self.OPTIONS)))
        check(self.lib.oa_plan_set_option(self.plan, self.OPTIONS[name], int(value)))
        # remembered: options belong to the plan, and Estimator.fork() builds new plans for the lanes of a multi-stream run
        self._options = dict(getattr(self, "_options", {}), **{name: int(value)})

    def copy_options_to(self, other):
        for name, value in getattr(self, "_options", {}).items():
            other.set_option(name, value)

    def release_pools(self):
        """free the plan-owned pools of the multi-map entries (oa_lens_maps, oa_qe_mv, oa_qe_tt_splits, oa_mc_run)"""
        check(self.lib.oa_plan_release_pools(self.plan))

    def set_laxes(self, ly, lx):
        ly = np.ascontiguousarray(ly, dtype=np.float64)
        lx = np.ascontiguousarray(lx, dtype=np.float64)
        if ly.shape != (self.ny,) or lx.shape != (self.nx,):
            raise ValueError("laxes must have shapes (ny,), (nx,)")
        check(self.lib.oa_plan_set_laxes(self.plan, ly.ctypes.data_as(ctypes.c_void_p), lx.ctypes.data_as(ctypes.c_void_p)))
        self._laxes = (ly, lx)

    def qe_legs(self, kX, kY, FG, FH, phase_g=0, phase_h=0, h_times_i=False, out=None):
        self._chk(kX, "hc"); self._chk(kY, "hc"); self._chk(FG, "hcreal"); self._chk(FH, "hcreal")
        if out is None:
            out = (self.hc(), self.hc(), self.hc())
        Gx, Gy, H = out
        _dirty(Gx); _dirty(Gy); _dirty(H)
        check(self.lib.oa_qe_legs(self.plan, _ptr(kX), _ptr(kY), _ptr(FG), _ptr(FH), _ptr(Gx), _ptr(Gy), _ptr(H),
                                  int(phase_g), int(phase_h), 1 if h_times_i else 0, _stream()))
        return Gx, Gy, H

    def qe_div(self, Px, Py, Fnorm, out=None, accumulate=False):
        self._chk(Px, "hc"); self._chk(Py, "hc"); self._chk(Fnorm, "hcreal")
        out = self.hc() if out is None else _dirty(self._chk(out, "hc"))
        check(self.lib.oa_qe_div(self.plan, _ptr(Px), _ptr(Py), _ptr(Fnorm), _ptr(out), 1 if accumulate else 0, _stream()))
        return out

    # ---- flat-sky lensing op ---------------------------------------------------------------
    def lens_split(self, alpha, step):
        self._chk(alpha, "real")
        shift = torch.empty(alpha.shape, dtype=torch.int32, device=self.device)
        delta = torch.empty_like(alpha)
        check(self.lib.oa_lens_split(self.code, _ptr(alpha), float(step), _ptr(shift), _ptr(delta), alpha.numel(), _stream()))
        return shift, delta

    def lens_gather(self, src, sx, sy, dx, dy, px, py, coef, out, accumulate):
        self._chk(src, "real"); _dirty(self._chk(out, "real")); self._chk(dx, "real"); self._chk(dy, "real")
        if sx.dtype != torch.int32 or sy.dtype != torch.int32 or sx.numel() != src.numel() or sy.numel() != src.numel():
            raise ValueError("lens_gather: shifts must be int32 planes")
        check(self.lib.oa_lens_gather(self.plan, _ptr(src), _ptr(sx), _ptr(sy), _ptr(dx), _ptr(dy), int(px), int(py),
                                      float(coef), _ptr(out), 1 if accumulate else 0, _stream()))
        return out

    # ---- binning -----------------------------------------------------------------------
    def digitize(self, x64, edges64):
        return dev_digitize(x64, edges64)

    def modl_digitize(self, edges64, half=False, want_modl=False):
        if self._laxes is None:
            raise RuntimeError("set_laxes first")
        ly = torch.as_tensor(self._laxes[0], device=self.device)
        lx = torch.as_tensor(self._laxes[1], device=self.device)
        pitch, width = (self.kp, self.nxh + 1) if half else (self.nx, self.nx)
        ids = torch.empty((self.ny, pitch), dtype=torch.int32, device=self.device)
        modl = torch.empty((self.ny, pitch), dtype=torch.float64, device=self.device) if want_modl else None
        check(self.lib.oa_modl_digitize(_ptr(ly), _ptr(lx), self.ny, self.nx, pitch, width, _ptr(edges64),
                                        edges64.numel(), _ptr(ids), _ptr(modl), _stream()))
        return (ids, modl) if want_modl else ids

    def bin(self, data, ids, nids, weights=None, aux=None, mode=0, skip_nan=False, herm=False):
        return dev_bin(data, ids, nids, weights=weights, aux=aux, mode=mode, skip_nan=skip_nan,
                       herm_pitch=self.kp if herm else 0, herm_nxh=self.nxh if herm else -1)

    def bin_power(self, k1, k2, norm, ids, nids, herm=True, active_cols=0, active_rows=0):
        return dev_bin_power(k1, k2, norm, ids, nids, herm_pitch=self.kp if herm else 0, herm_nxh=self.nxh if herm else -1,
                             active_cols=active_cols if herm else 0, active_rows=active_rows if herm else 0)

    # ---- random fields / accumulators ---------------------------------------------------
    def grf_hc(self, seed, stream_id, covsqrt_hc=None, out=None, width=0, rband=0):
        """Hermitian-consistent Gaussian draw on the hc grid (Philox key = (seed, stream_id)).  width / rband > 0: only
        the active region (columns < width, rows |ky index| < rband) is drawn -- the same values as the full draw
        there, the rest of ``out`` untouched (zero for a fresh plane)."""
        if covsqrt_hc is not None:
            self._chk(covsqrt_hc, "hcreal")
        if out is None:
            out = self.hc_fresh() if (width <= 0 and rband <= 0) else self.hc()
        else:
            _dirty(self._chk(out, "hc"))
        check(self.lib.oa_grf_hc_band(self.plan, int(seed), int(stream_id), _ptr(covsqrt_hc), _ptr(out), int(width), int(rband), _stream()))
        return out

    def grf_mix(self, seed, covsqrt_hc, rot=None, inputs=None, filt=None, scale=1.0, out=None, stream_id0=0):
        """``oa_grf_mix``: the draw of :meth:`grf_hc` for streams stream_id0 .. stream_id0 + n - 1, mixed by the n x n table of
        hc-real planes ``covsqrt_hc`` (None = zero block), rotated by ``rot`` = (c, s) on components 1, 2 and -- with
        ``inputs`` (n hc planes) -- added, times ``scale``, to rot(inputs * filt); one pass.  ``out``: n hc planes
        (may be the inputs); returns them as a list."""
        import ctypes
        n = len(covsqrt_hc)
        if not (1 <= n <= 3) or any(len(r) != n for r in covsqrt_hc):
            raise ValueError("grf_mix: covsqrt_hc must be an n x n table, 1 <= n <= 3")
        if rot is not None and n != 3:
            raise ValueError("grf_mix: the rotation acts on components 1, 2 of three")
        if filt is not None and inputs is None:
            raise ValueError("grf_mix: a filter without input planes")
        for r in covsqrt_hc:
            for c in r:
                if c is not None:
                    self._chk(c, "hcreal")
        if rot is not None:
            self._chk(rot[0], "hcreal"); self._chk(rot[1], "hcreal")
        if filt is not None:
            self._chk(filt, "hcreal")
        if inputs is not None:
            if len(inputs) != n:
                raise ValueError("grf_mix: %d input planes for %d components" % (len(inputs), n))
            for k in inputs:
                self._chk(k, "hc")
        if out is None:                    # the kernel writes the row padding too: no fill
            t = torch.empty((n, self.ny, self.kp), dtype=self.cdt, device=self.device)
            out = [t[i] for i in range(n)]
        else:
            if len(out) != n:
                raise ValueError("grf_mix: %d output planes for %d components" % (len(out), n))
            for k in out:
                _dirty(self._chk(k, "hc"))
        PT = ctypes.c_void_p * (n * n)
        cs = PT(*[_ptr(covsqrt_hc[i][j]) for i in range(n) for j in range(n)])
        P3 = ctypes.c_void_p * n
        outs = P3(*[_ptr(k) for k in out])
        ins = P3(*[_ptr(k) for k in inputs]) if inputs is not None else None
        check(self.lib.oa_grf_mix(self.plan, int(seed), int(stream_id0), n, cs, _ptr(rot[0]) if rot is not None else None,
                                  _ptr(rot[1]) if rot is not None else None, ins, _ptr(filt), float(scale), outs, _stream()))
        return list(out)

    def randn(self, seed, stream_id, shape=None):
        out = torch.empty(shape if shape is not None else (self.ny, self.nx), dtype=self.rdt, device=self.device)
        check(self.lib.oa_randn(self.code, int(seed), int(stream_id), _ptr(out), out.numel(), _stream()))
        return out
