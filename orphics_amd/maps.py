"""orphics.maps hot-path surface on MI355X: FourierCalc, MapGen, filter_map,
gauss_beam, mask_kspace, tapers, binned_power.

Signatures mirror /root/reference/orphics/maps.py (cited per function).  The
FFT / per-mode arithmetic runs in the HIP kernels behind include/orphics_amd.h;
there is no CPU fallback.

Data conventions
----------------
* NumPy in -> NumPy out, full-plane float64/complex128 like the reference
  (precision of the plan follows the input dtype: float32 maps use the f32
  kernels, float64 the f64 kernels).
* CUDA tensors in -> CUDA tensors out (full-plane).
* ``FourierCalc(..., layout="half")``: Fourier-space results are
  :class:`~orphics_amd.stats.HalfPlane` objects (non-redundant half plane of a
  real field, the device-native fast path); every method also accepts them.
"""
import numpy as np

from .geometry import FlatGeometry, as_geometry, rect_geometry  # noqa: F401
from .stats import HalfPlane


def _torch():
    import torch
    return torch


def _is_tensor(x):
    torch = _torch()
    return isinstance(x, torch.Tensor)


def _engine(shape, prec):
    from .engine import Engine
    Ny, Nx = shape[-2:]
    if Ny < 32 or Nx < 32 or Ny % 2 or Nx % 2:
        raise NotImplementedError("orphics_amd FFTs need even map sides >= 32 (powers of two for the fast fused "
                                  "kernels, any other even side through the chirp-z path), got %dx%d" % (Ny, Nx))
    return Engine.get(Ny, Nx, prec)


def _prec(x):
    from .engine import precision_of
    if isinstance(x, HalfPlane):
        return x.eng.prec
    return precision_of(x)


def _ret(t, like):
    """Return ``t`` (device tensor) in the container kind of ``like``."""
    if isinstance(like, np.ndarray) or not (_is_tensor(like) or isinstance(like, HalfPlane)):
        return t.cpu().numpy()
    return t


def _planes(x):
    """Flatten leading dims: (..., Ny, Nx) -> list of 2-D slices + lead shape."""
    lead = tuple(x.shape[:-2])
    flat = x.reshape((-1,) + tuple(x.shape[-2:]))
    return [flat[i] for i in range(flat.shape[0])], lead


def gauss_beam(ell, fwhm):
    """maps.py:1925-1927."""
    tht_fwhm = np.deg2rad(fwhm / 60.)
    return np.exp(-(tht_fwhm ** 2.) * (ell ** 2.) / (16. * np.log(2.)))


def sigma_from_fwhm(fwhm):
    return fwhm / 2. / np.sqrt(2. * np.log(2.))


def fwhm_from_sigma(sigma):
    return 2. * np.sqrt(2. * np.log(2.)) * sigma


def mask_kspace(shape, wcs, lxcut=None, lycut=None, lmin=None, lmax=None):
    """0/1 (int64) mode mask of maps.py:1936-1948.  A mode SURVIVES only under strict inequalities: lmin < ell < lmax,
    |lx| >= lxcut, |ly| >= lycut; cuts that are None do not apply."""
    geom = as_geometry(shape, wcs)
    keep = np.ones(tuple(shape[-2:]), dtype=bool)
    if lmin is not None or lmax is not None:
        ell = geom.modlmap()
        if lmin is not None:
            keep &= ell > lmin
        if lmax is not None:
            keep &= ell < lmax
    if lxcut is not None or lycut is not None:
        ly, lx = geom.laxes()
        if lxcut is not None:
            keep &= (np.abs(lx) >= lxcut)[None, :]
        if lycut is not None:
            keep &= (np.abs(ly) >= lycut)[:, None]
    return keep.astype(int)


def _edge_taper(n, width, pad):
    """The two one-sided raised-cosine factors of an n-pixel axis (leading edge, trailing edge): 1 outside their
    ranges `x <= width + pad` and `x >= n - 1 - width - pad`, (1 - cos(pi d / width)) / 2 inside, d = distance from the
    padded edge (negative inside the pad, as in the reference, which zeroes the pad afterwards)."""
    x = np.arange(n)
    lead, trail = np.ones(n), np.ones(n)
    if width > 0:
        ramp = lambda d: 1. / 2 * (1 - np.cos(-np.pi * d / width))       # noqa: E731  (the reference's rounding)
        lo = x <= width + pad
        hi = x >= (n - 1) - width - pad
        lead[lo] = ramp((x - pad).astype(float)[lo])
        trail[hi] = ramp(((n - 1) - x - pad).astype(float)[hi])
    return lead, trail, (lo if width > 0 else np.zeros(n, bool)), (hi if width > 0 else np.zeros(n, bool))


def cosine_window(Ny, Nx, lenApodY=30, lenApodX=30, padY=0, padX=0):
    """Separable raised-cosine apodisation with zeroed pads (maps.py:1891-1920), built from two 1-D profiles.
    Faithful to the reference where the two edge ramps of an axis overlap (narrow maps): along x the trailing ramp
    REPLACES the leading one, along y the two MULTIPLY."""
    xl, xt, _, xhi = _edge_taper(Nx, lenApodX, padX)
    yl, yt, _, _ = _edge_taper(Ny, lenApodY, padY)
    wx = np.where(xhi, xt, xl)
    win = (wx[None, :] * yl[:, None]) * yt[:, None]
    win[:padY] = 0
    win[Ny - padY:] = 0
    win[:, :padX] = 0
    win[:, Nx - padX:] = 0
    return win


def get_taper(shape, wcs=None, taper_percent=12.0, pad_percent=3.0, weight=None):
    """maps.py:1873-1878: returns (taper, mean(taper^2))."""
    Ny, Nx = shape[-2:]
    if weight is None:
        weight = np.ones(shape[-2:])
    n = min(Ny, Nx)
    taper = cosine_window(Ny, Nx, lenApodY=int(taper_percent * n / 100.), lenApodX=int(taper_percent * n / 100.),
                          padY=int(pad_percent * n / 100.), padX=int(pad_percent * n / 100.)) * weight
    return taper, np.mean(taper ** 2.)


def get_taper_deg(shape, wcs, taper_width_degrees=1.0, pad_width_degrees=0., weight=None, only_y=False):
    """maps.py:1880-1888."""
    Ny, Nx = shape[-2:]
    if weight is None:
        weight = np.ones(shape[-2:])
    res = abs(as_geometry(shape, wcs).step_y)
    pix_apod = int(taper_width_degrees * np.pi / 180. / res)
    pix_pad = int(pad_width_degrees * np.pi / 180. / res)
    taper = cosine_window(Ny, Nx, lenApodY=pix_apod, lenApodX=pix_apod if not only_y else 0, padY=pix_pad,
                          padX=pix_pad if not only_y else 0) * weight
    return taper, np.mean(taper ** 2.)


def queb_rotmat(lmap, inverse=False, iau=False, spin=2):
    """pixell enmap.queb_rotmat as used by maps.py:1607."""
    sgn = 1 if iau else -1
    a = sgn * spin * np.arctan2(-lmap[1], lmap[0])
    c, s = np.cos(a), np.sin(a)
    if inverse:
        s = -s
    return np.array([[c, -s], [s, c]])


class FourierCalc(object):
    """maps.py:1594-1677."""

    def __init__(self, shape, wcs, iau=False, layout="full"):
        self.shape = tuple(shape)
        self.wcs = wcs
        self.geom = as_geometry(shape, wcs)
        self.layout = layout
        assert layout in ("full", "half")
        self.normfact = self.geom.area / np.prod(self.shape[-2:]) ** 2.
        self.iau = iau
        self._rot_dev = {}
        if len(shape) > 2 and shape[-3] > 1:
            self.rot = queb_rotmat(self.geom.lmap(), iau=iau)

    # ---- helpers --------------------------------------------------------------
    def _eng(self, x):
        return _engine(self.shape, _prec(x))

    def _rot_planes(self, eng, half):
        key = (eng.prec, half)
        if key not in self._rot_dev:
            c = eng.to_real(self.rot[0, 0])
            s = eng.to_real(self.rot[1, 0])
            if half:
                c, s = eng.fullreal_to_hc(c), eng.fullreal_to_hc(s)
            self._rot_dev[key] = (c, s)
        return self._rot_dev[key]

    def _fft_real_planes(self, emap, eng, scale):
        """list of hc tensors, one per 2-D slice of the real map ``emap``."""
        torch = _torch()
        if isinstance(emap, np.ndarray) and np.iscomplexobj(emap) or (_is_tensor(emap) and emap.is_complex()):
            raise TypeError("complex maps go through the C2C path")
        x = eng.to_real(emap)
        planes, lead = _planes(x)
        return [eng.rfft(p.contiguous(), scale=scale) for p in planes], lead

    def _to_half(self, k, eng):
        """Any k-space input -> HalfPlane (assumes Hermitian symmetry for full inputs)."""
        if isinstance(k, HalfPlane):
            return k
        torch = _torch()
        z = eng.to_complex(k)
        planes, lead = _planes(z)
        hs = [eng.full_to_hc(p.contiguous()) for p in planes]
        return HalfPlane(torch.stack(hs).reshape(lead + (eng.ny, eng.kp)), eng)

    def _to_full(self, k, eng):
        if isinstance(k, HalfPlane):
            return k.full()
        return eng.to_complex(k)

    def _wrap_k(self, hs, lead, eng, like):
        """hc plane list -> output in the configured layout / container kind."""
        torch = _torch()
        t = torch.stack(hs).reshape(lead + (eng.ny, eng.kp))
        hp = HalfPlane(t, eng)
        if self.layout == "half":
            return hp
        return _ret(hp.full(), like)

    # ---- reference API ----------------------------------------------------------
    def iqu2teb(self, emap, nthread=0, normalize=True, rot=True):
        """maps.py:1609-1617: 2-D FFT (enmap.fft normalize=True is 1/sqrt(Npix)),
        then per-mode Q,U -> E,B rotation of the last two components."""
        torch = _torch()
        eng = self._eng(emap)
        scale = 1.0 / np.sqrt(eng.npix) if normalize else 1.0
        hs, lead = self._fft_real_planes(emap, eng, scale)
        dorot = len(lead) >= 1 and lead[-1] > 1 and rot
        if not dorot:
            return self._wrap_k(hs, lead, eng, emap)
        ncomp = lead[-1]
        if self.layout == "half":
            # Hermitian convention at the self-conjugate Nyquist modes
            c, s = self._rot_planes(eng, True)
            for b in range(0, len(hs), ncomp):
                hs[b + ncomp - 2], hs[b + ncomp - 1] = eng.rot2(c, s, hs[b + ncomp - 2], hs[b + ncomp - 1])
            return self._wrap_k(hs, lead, eng, emap)
        # full layout: rotate on the full plane with the full-plane matrix, exactly like
        # enmap.map_mul(self.rot, ...) (maps.py:1614-1615), Nyquist modes included
        c, s = self._rot_planes(eng, False)
        fs = [eng.hc_to_full(h) for h in hs]
        for b in range(0, len(fs), ncomp):
            fs[b + ncomp - 2], fs[b + ncomp - 1] = eng.rot2(c, s, fs[b + ncomp - 2], fs[b + ncomp - 1])
        return _ret(torch.stack(fs).reshape(lead + (eng.ny, eng.nx)), emap)

    def f2power(self, kmap1, kmap2, pixel_units=False):
        """maps.py:1620-1624: Re(conj(k1) k2) * norm."""
        torch = _torch()
        norm = 1. if pixel_units else self.normfact
        if isinstance(kmap1, HalfPlane) or isinstance(kmap2, HalfPlane):
            eng = kmap1.eng if isinstance(kmap1, HalfPlane) else kmap2.eng
            a, b = self._to_half(kmap1, eng), self._to_half(kmap2, eng)
            return HalfPlane(eng.f2power(a.t.contiguous(), b.t.contiguous(), norm), eng)
        eng = self._eng(kmap1)
        a, b = eng.to_complex(kmap1), eng.to_complex(kmap2)
        return _ret(eng.f2power(a, b, norm), kmap1)

    def f1power(self, map1, kmap2, pixel_units=False, nthread=0):
        """maps.py:1626-1630."""
        kmap1 = self.iqu2teb(map1, nthread, normalize=False)
        return self.f2power(kmap1, kmap2, pixel_units), kmap1

    def ifft(self, kmap):
        """maps.py:1632-1633: inverse C2C divided by Npix; returns complex like
        the reference (HalfPlane input -> real map via C2R, the imaginary part is
        identically zero there)."""
        torch = _torch()
        if isinstance(kmap, HalfPlane):
            eng = kmap.eng
            planes, lead = _planes(kmap.t)
            outs = [eng.irfft(p.contiguous()) for p in planes]
            return torch.stack(outs).reshape(lead + (eng.ny, eng.nx))
        eng = self._eng(kmap)
        z = eng.to_complex(kmap)
        planes, lead = _planes(z)
        outs = [eng.cfft(p.contiguous(), inverse=True, scale=1.0 / eng.npix) for p in planes]
        return _ret(torch.stack(outs).reshape(lead + (eng.ny, eng.nx)), kmap)

    def fft(self, emap):
        """maps.py:1635-1636: unnormalised forward transform."""
        torch = _torch()
        cplx = (isinstance(emap, np.ndarray) and np.iscomplexobj(emap)) or (_is_tensor(emap) and emap.is_complex())
        if cplx:
            eng = self._eng(emap)
            z = eng.to_complex(emap)
            planes, lead = _planes(z)
            outs = [eng.cfft(p.contiguous()) for p in planes]
            return _ret(torch.stack(outs).reshape(lead + (eng.ny, eng.nx)), emap)
        eng = self._eng(emap)
        hs, lead = self._fft_real_planes(emap, eng, 1.0)
        return self._wrap_k(hs, lead, eng, emap)

    def power2d(self, emap=None, emap2=None, nthread=0, pixel_units=False, skip_cross=False, rot=True, kmap=None,
                kmap2=None, dtype=None):
        """maps.py:1639-1677."""
        torch = _torch()
        if kmap is not None:
            lteb1 = kmap
            ndim = len(kmap.shape)
            if ndim > 2:
                ncomp = kmap.shape[-3]
        else:
            lteb1 = self.iqu2teb(emap, nthread, normalize=False, rot=rot)
            ndim = emap.ndim
            if ndim > 2:
                ncomp = emap.shape[-3]
        if kmap2 is not None:
            lteb2 = kmap2
        else:
            lteb2 = self.iqu2teb(emap2, nthread, normalize=False, rot=rot) if emap2 is not None else lteb1
        assert tuple(lteb1.shape) == tuple(lteb2.shape)
        if ndim > 2 and ncomp > 1:
            pw = {}
            for i in range(ncomp):
                pw[(i, i)] = self.f2power(lteb1[i], lteb2[i], pixel_units)
            if not skip_cross:
                for i in range(ncomp):
                    for j in range(i + 1, ncomp):
                        pw[(i, j)] = self.f2power(lteb1[i], lteb2[j], pixel_units)
                        pw[(j, i)] = pw[(i, j)]
            first = pw[(0, 0)]
            if isinstance(first, HalfPlane):
                eng = first.eng
                ret = torch.zeros((ncomp, ncomp, eng.ny, eng.kp), dtype=first.t.dtype, device=first.t.device)
                for (i, j), v in pw.items():
                    ret[i, j] = v.t
                retpow = HalfPlane(ret, eng)
            elif isinstance(first, np.ndarray):
                retpow = np.zeros((ncomp, ncomp) + first.shape[-2:], dtype=dtype)
                for (i, j), v in pw.items():
                    retpow[i, j] = v
            else:
                retpow = torch.zeros((ncomp, ncomp) + tuple(first.shape[-2:]), dtype=first.dtype, device=first.device)
                for (i, j), v in pw.items():
                    retpow[i, j] = v
            return retpow, lteb1, lteb2
        if len(lteb1.shape) > 2:
            lteb1 = lteb1[0]
        if len(lteb2.shape) > 2:
            lteb2 = lteb2[0]
        p2d = self.f2power(lteb1, lteb2, pixel_units)
        return p2d, lteb1, lteb2


class HalfFilter(object):
    """An even-symmetric real k-space filter resident on the device in the half-plane layout (:func:`prepare_filter`)."""

    def __init__(self, t, eng):
        self.t, self.eng = t, eng


def prepare_filter(shape, kfilter, dtype="f32"):
    """Upload a full-plane (Ny, Nx) even-symmetric real filter once; the result can be passed to :func:`filter_map` any
    number of times (a Monte-Carlo loop applying the same beam to every realisation: ``FlatLensingSims``)."""
    torch = _torch()
    eng = _engine(tuple(shape), dtype)
    fdev = eng.to_real(np.broadcast_to(np.asarray(kfilter, dtype=np.float64), tuple(shape)[-2:]))
    flipped = torch.roll(torch.flip(fdev, dims=(0, 1)), shifts=(1, 1), dims=(0, 1))
    if not torch.equal(fdev, flipped):
        raise ValueError("prepare_filter: the filter is not even-symmetric (F(-l) != F(l)); pass the array to filter_map instead")
    return HalfFilter(eng.fullreal_to_hc(fdev), eng)


def filter_map(imap, kfilter):
    """maps.py:1922-1923: Re(IFFT(FFT(m) * F)), IFFT / Npix.  ``kfilter`` is a
    full-plane (Ny,Nx) array: an even-symmetric real (or int) filter keeps the
    result exactly real and allows the half-plane R2C/C2R path, a general real
    or a complex filter takes the C2C path like the reference."""
    torch = _torch()
    from .engine import precision_of
    shape = tuple(imap.shape)
    eng = _engine(shape, precision_of(imap))
    x = eng.to_real(imap)
    f = kfilter
    if (isinstance(f, np.ndarray) and np.iscomplexobj(f)) or (_is_tensor(f) and f.is_complex()):
        # complex filter (a phase: shifts, derivative operators): full-plane C2C like the reference
        fc = eng.to_complex(np.broadcast_to(np.asarray(f), shape[-2:]) if not _is_tensor(f) else f)
        planes, lead = _planes(x)
        outs = []
        for p in planes:
            k = eng.hc_to_full(eng.rfft(p.contiguous()))
            k = eng.cmul(k, fc, out=k)
            outs.append(torch.real(eng.cfft(k, inverse=True, scale=1.0 / eng.npix)).contiguous())
        return _ret(torch.stack(outs).reshape(lead + (eng.ny, eng.nx)), imap)
    if isinstance(f, HalfFilter):
        # a filter prepared once with prepare_filter(): no conversion, no symmetry test per call
        if f.eng is not eng:
            raise ValueError("filter_map: the prepared filter belongs to another geometry / precision")
        planes, lead = _planes(x)
        outs = [eng.irfft(eng.cmul_real(eng.rfft(p.contiguous()), f.t)) for p in planes]
        return _ret(torch.stack(outs).reshape(lead + (eng.ny, eng.nx)), imap)
    fdev = eng.to_real(np.broadcast_to(np.asarray(f, dtype=np.float64), shape[-2:]) if not _is_tensor(f) else f)
    planes, lead = _planes(x)
    # even-symmetry test decides the fast path
    flipped = torch.roll(torch.flip(fdev, dims=(0, 1)), shifts=(1, 1), dims=(0, 1))
    if torch.equal(fdev, flipped):
        fh = eng.fullreal_to_hc(fdev)
        outs = []
        for p in planes:
            k = eng.rfft(p.contiguous())
            k = eng.cmul_real(k, fh, out=k)
            outs.append(eng.irfft(k))
    else:
        outs = []
        for p in planes:
            k = eng.hc_to_full(eng.rfft(p.contiguous()))
            k = eng.cmul_real(k, fdev, out=k)
            outs.append(torch.real(eng.cfft(k, inverse=True, scale=1.0 / eng.npix)).contiguous())
    return _ret(torch.stack(outs).reshape(lead + (eng.ny, eng.nx)), imap)


def multi_pow(cov, exp):
    """pixell enmap.multi_pow (host, one-off per MapGen)."""
    nc = cov.shape[0]
    if nc == 1:
        return np.abs(cov) ** exp
    m = np.moveaxis(cov.reshape(nc, nc, -1), -1, 0)
    w, v = np.linalg.eigh(m)
    w = np.where(w > 0, w, 0.0) ** exp
    res = np.einsum("pab,pb,pcb->pac", v, w, v)
    return np.moveaxis(res, 0, -1).reshape(cov.shape)


def smooth_spectrum(ps, width):
    """Gaussian smoothing of 1-D spectra (..., nl) along ell with the 2-D mode density (~ ell) as weight:
    ps_s(l) = sum_l' K(l - l') l' ps(l') / sum_l' K(l - l') l', K = exp(-(dl / width)^2 / 2).
    Stands in for pixell ``enmap.smooth_spectrum(kernel="gauss", weight="mode")`` inside ``spec2flat``
    [NOT IN SNAPSHOT: pixell is not vendored, so the exact kernel / weight are a design choice here]: it
    approximates averaging the spectrum over the sub-grid modes each 2-D pixel of width delta-ell stands for."""
    ps = np.asarray(ps, dtype=np.float64)
    nl = ps.shape[-1]
    if width <= 0 or nl < 2:
        return ps
    half = int(min(nl - 1, np.ceil(5 * width)))
    dl = np.arange(-half, half + 1, dtype=np.float64)
    K = np.exp(-0.5 * (dl / width) ** 2)
    ell = np.arange(nl, dtype=np.float64)
    w = np.maximum(ell, 0.5)                      # ell = 0 keeps a small weight instead of dropping out
    flat = ps.reshape(-1, nl)
    num = np.stack([np.convolve(f * w, K, mode="same") for f in flat])
    den = np.convolve(w, K, mode="same")
    return (num / den).reshape(ps.shape)


def spec2flat(shape, wcs, cov, exp=1.0, mode="constant", smooth="auto"):
    """Isotropic (ncomp,ncomp,nl) spectra sampled at ell = 0..nl-1 -> per-mode (ncomp,ncomp,Ny,Nx) planes in
    PIXEL units (x Npix/area), raised to the matrix power ``exp`` -- what ``MapGen`` needs from
    ``enmap.spec2flat(shape, wcs, cov, 0.5, mode="constant", smooth=smooth)`` (maps.py:1573).
    Order of operations [NOT IN SNAPSHOT -- pixell's public behaviour as recalled, documented so that it can be
    checked against a pixell installation]: (1) ``smooth="auto"``: smooth the 1-D spectra over a Gaussian of width
    0.5 (delta-ell_y + delta-ell_x) / 3.41 (:func:`smooth_spectrum`); a number = that width; 0 / None = none;
    (2) scale by Npix / area; (3) per-ell symmetric matrix power; non-finite -> 0; (4) LINEAR interpolation onto
    |ell| of every 2-D mode, ``mode="constant"``: 0 beyond the table."""
    geom = as_geometry(shape, wcs)
    cov = np.asarray(cov, dtype=np.float64)
    if cov.ndim == 1:
        cov = cov[None, None]
    assert cov.ndim == 3 and cov.shape[0] == cov.shape[1], "cov must be (ncomp,ncomp,nl)"
    ml = geom.modlmap()
    if smooth == "auto":
        ly, lx = geom.laxes()
        smooth = 0.5 * (abs(ly[1] - ly[0]) + abs(lx[1] - lx[0])) / 3.41
    if smooth:
        cov = smooth_spectrum(cov, float(smooth))
    Ny, Nx = geom.shape[-2:]
    cov = cov * (Ny * Nx / geom.area)
    if exp != 1.0:
        cov = multi_pow(cov, exp)
    cov = np.where(np.isfinite(cov), cov, 0.0)
    nc, nl = cov.shape[0], cov.shape[-1]
    ell = np.arange(nl, dtype=np.float64)
    out = np.empty((nc, nc) + ml.shape)
    fill = 0.0 if mode == "constant" else None
    for i in range(nc):
        for j in range(nc):
            out[i, j] = np.interp(ml, ell, cov[i, j], left=cov[i, j, 0], right=(fill if fill is not None else cov[i, j, -1]))
    return out


def Ny_Nx_not_divisible(shape, ndown):
    Ny, Nx = tuple(shape)[-2:]
    nd = np.array(ndown).ravel()
    if nd.size == 1:
        other = int(nd[0] * max(Ny, Nx) * 1. / min(Ny, Nx))
        fy, fx = (other, int(nd[0])) if Ny > Nx else (int(nd[0]), other)
    else:
        fy, fx = int(nd[0]), int(nd[1])
    return fy < 1 or fx < 1 or Ny % max(fy, 1) != 0 or Nx % max(fx, 1) != 0


def downsample_power(shape, wcs, cov, ndown=16, order=0, exp=None, fftshift=True, fft=False, logfunc=lambda x: x,
                     ilogfunc=lambda x: x, fft_up=False):
    """maps.py:1501-1550: smooth a 2-D power spectrum (..., Ny, Nx) by averaging it over blocks of ``ndown`` Fourier
    pixels and interpolating the block means back onto the full grid (a noise-model aid; one-off host arithmetic, as
    in the reference).  ``ndown``: one factor (scaled by the aspect ratio for the longer axis, maps.py:1512-1518) or a
    (ndown_y, ndown_x) pair; ``exp``: per-mode matrix power of the block means (``MapGen`` passes 0.5).

    Steps of the default path, as the reference composes them from pixell calls [NOT IN SNAPSHOT: pixell's public
    behaviour as recalled -- ``enmap.downgrade`` = mean over whole blocks, a trailing partial block dropped;
    ``ndmap.at(pix, unit="pix", order=order, mask_nan=False)`` = spline interpolation of that order with zeros
    outside the coarse grid]: fftshift -> block means -> matrix power -> sample the coarse plane at
    (y / ndown_y, x / ndown_x) for every fine pixel (y, x) -> inverse fftshift.  The ``fft`` / ``fft_up`` variants
    resample with ``pixell.resample.resample_fft`` and are not provided."""
    from scipy import ndimage
    if np.all(np.asarray(ndown) < 1):
        return cov
    if fft or fft_up:
        raise NotImplementedError("downsample_power(fft=True / fft_up=True): Fourier resampling (pixell.resample) is outside the hot path")
    if order > 0 or Ny_Nx_not_divisible(shape, ndown):
        import warnings
        warnings.warn("downsample_power: the block-mean / order-%d sampling step restates pixell's enmap.downgrade and "
                      "ndmap.at from their documented behaviour (pixell is absent: parity unpinned, SURVEY.md F3); at the "
                      "plane edges and for sides that are not multiples of ndown the result may differ from the reference's"
                      % order, stacklevel=2)
    Ny, Nx = tuple(shape)[-2:]
    nd = np.array(ndown).ravel()
    if nd.size == 1:
        other = int(nd[0] * max(Ny, Nx) * 1. / min(Ny, Nx))
        fy, fx = (other, int(nd[0])) if Ny > Nx else (int(nd[0]), other)
    else:
        assert nd.size == 2
        fy, fx = int(nd[0]), int(nd[1])
    plane = logfunc(np.asarray(cov, dtype=np.float64))
    lead = plane.shape[:-2]
    if fftshift:
        plane = np.fft.fftshift(plane, axes=(-2, -1))
    cy, cx = Ny // fy, Nx // fx
    blocks = plane[..., :cy * fy, :cx * fx].reshape(lead + (cy, fy, cx, fx)).mean(axis=(-3, -1))
    if exp is not None:
        blocks = multi_pow(blocks, exp)
    yy, xx = np.meshgrid(np.arange(Ny) / float(fy), np.arange(Nx) / float(fx), indexing="ij")
    fine = np.empty(lead + (Ny, Nx))
    for idx in np.ndindex(*lead):
        fine[idx] = ndimage.map_coordinates(blocks[idx], [yy, xx], order=order, mode="constant", cval=0.0)
    if fftshift:
        fine = np.fft.ifftshift(fine, axes=(-2, -1))
    return ilogfunc(fine)


class MapGen(object):
    """maps.py:1553-1587.  ``cov`` is the 4-D per-mode covariance (ncomp,ncomp,Ny,Nx) or the 3-D isotropic
    form (ncomp,ncomp,lmax) sampled at integer ell (expanded with :func:`spec2flat`, maps.py:1573).
    ``ndown``: the 4-D covariance is block-averaged and re-interpolated before the square root
    (:func:`downsample_power`, maps.py:1501-1550, 1568-1569).

    ``get_map`` draws on the device (Philox; the reference's global
    Mersenne-Twister stream cannot and need not be reproduced);
    ``get_map_from_rand`` applies the reference arithmetic to caller-supplied
    white noise (the parity entry point)."""

    def __init__(self, shape, wcs, cov=None, covsqrt=None, pixel_units=False, smooth="auto", ndown=None, order=1,
                 dtype="f32"):
        self.shape = tuple(shape)
        self.wcs = wcs
        self.geom = as_geometry(shape, wcs)
        self.prec = dtype
        if covsqrt is not None:
            self.covsqrt = np.asarray(covsqrt)
        else:
            assert cov is not None and cov.ndim >= 3, \
                "Power spectra have to be of shape (ncomp,ncomp,lmax) or (ncomp,ncomp,Ny,Nx)."
            cov = np.asarray(cov)
            if cov.ndim == 4:
                if not pixel_units:
                    cov = cov * np.prod(self.shape[-2:]) / self.geom.area
                if ndown:
                    self.covsqrt = downsample_power(self.shape, self.geom, cov, ndown, order, exp=0.5)     # maps.py:1568-1569
                else:
                    self.covsqrt = multi_pow(cov, 0.5)
            elif cov.ndim == 3:
                # maps.py:1573 (pixel_units plays no role on this branch in the reference either)
                self.covsqrt = spec2flat(self.shape, self.geom, cov, 0.5, mode="constant", smooth=smooth)
            else:
                raise AssertionError("Power spectra have to be of shape (ncomp,ncomp,lmax) or (ncomp,ncomp,Ny,Nx).")
        self.ncomp = self.covsqrt.shape[0]
        self._cs_dev = {}
        self._rot_dev = {}
        self._calls = 0
        # which (i, j) blocks of the square root are non-zero: decided ONCE (a per-call np.any over (nc, nc) full planes
        # was 1.4 of the 1.45 s a 4096^2 lensed simulation took on the host, profiles/r03f_lensloop.txt)
        self._nz = [[bool(np.any(self.covsqrt[i, j])) for j in range(self.ncomp)] for i in range(self.ncomp)]

    def _covsqrt_hc(self, eng):
        key = eng.prec
        if key not in self._cs_dev:
            nc = self.ncomp
            self._cs_dev[key] = [[eng.fullreal_to_hc(eng.to_real(self.covsqrt[i, j])) for j in range(nc)] for i in range(nc)]
        return self._cs_dev[key]

    def get_map(self, seed=None, scalar=False, iau=False, real=False, harm=False):
        """maps.py:1576-1587 semantics with an on-device Hermitian draw:
        C2R of covsqrt * (Hermitian unit white noise) with the unitary scale is
        statistically identical to ``enmap.ifft(covsqrt * rand_gauss_harm).real``.
        ``real=True`` (maps.py:1578): the white noise is drawn in MAP space (one N(0,1) per pixel, Philox) and
        transformed (unitary R2C) instead of being drawn mode by mode -- same statistics, the reference's other path."""
        torch = _torch()
        eng = _engine(self.shape, self.prec)
        seed = self._seed_int(seed)
        cs = self._covsqrt_hc(eng)
        nc = self.ncomp
        # a component that couples to no other (its row and column of covsqrt hold the diagonal entry only: white noise, kappa, the
        # B mode of the unlensed CMB) is drawn WITH its amplitude (the draw kernel multiplies by covsqrt): no white plane, no
        # separate multiply
        alone = [all((not self._nz[i][j]) and (not self._nz[j][i]) for j in range(nc) if j != i) and bool(self._nz[i][i]) for i in range(nc)]
        if real:
            white = [eng.rfft(eng.randn(seed, c), scale=1.0 / np.sqrt(eng.npix)) for c in range(nc)]
            alone = [False] * nc
        else:
            white = [None if alone[c] else eng.grf_hc(seed, c) for c in range(nc)]
        ks = []
        for i in range(nc):
            if alone[i]:
                ks.append(eng.grf_hc(seed, i, cs[i][i]))
                continue
            acc = None
            for j in range(nc):
                if not self._nz[i][j]:
                    continue
                term = eng.cmul_real(white[j], cs[i][j])
                acc = term if acc is None else acc + term
            ks.append(acc if acc is not None else eng.hc())
        if harm:
            return HalfPlane(torch.stack(ks) if len(self.shape) > 2 else ks[0], eng)
        if not scalar and nc == 3:
            # harm2map: E,B -> Q,U by the inverse rotation, then inverse FFT
            key = (eng.prec, bool(iau))
            if key not in self._rot_dev:          # device planes of the inverse rotation, made once per precision / convention
                rot = queb_rotmat(self.geom.lmap(), inverse=True, iau=iau)
                self._rot_dev[key] = (eng.fullreal_to_hc(eng.to_real(rot[0, 0])), eng.fullreal_to_hc(eng.to_real(rot[1, 0])))
            c, s = self._rot_dev[key]
            ks[1], ks[2] = eng.rot2(c, s, ks[1], ks[2])
        if len(self.shape) > 2:
            out = torch.empty((nc, eng.ny, eng.nx), dtype=eng.rdt, device=eng.device)     # one allocation, no stack copy
            for i, k in enumerate(ks):
                eng.irfft(k, scale=1.0 / np.sqrt(eng.npix), out=out[i])
            return out
        return eng.irfft(ks[0], scale=1.0 / np.sqrt(eng.npix))

    def _seed_int(self, seed):
        if seed is None:
            return int(np.random.randint(0, 2 ** 31 - 1))
        if isinstance(seed, (tuple, list)):
            return int(np.random.SeedSequence(list(seed)).generate_state(1, dtype=np.uint64)[0] >> 1)
        return int(seed)

    def draw_hc(self, seed=None, rot=None, inputs=None, filt=None, scale=1.0, out=None, iau=False):
        """``get_map(seed, harm=True)``'s transforms in ONE device pass (``oa_grf_mix``) -- the same white fields (Philox streams
        (seed, component)), mixed by covsqrt in the same order of operations -- with what usually follows folded in:
          rot="inverse" : harm2map's E, B -> Q, U rotation (maps.py:1584-1586) applied to the draw,
          inputs (+filt): the draw, times ``scale``, is ADDED to rot(inputs * filt) -- rot="forward": Q, U -> E, B
                          (FourierCalc.iqu2teb's rotation) -- i.e. beam x signal + noise in the T, E, B basis.
        Returns the (ncomp, Ny, kp) stacked hc tensor (the drawn transforms are unitary: rfft(map) = sqrt(Npix) x them)."""
        torch = _torch()
        eng = _engine(self.shape, self.prec)
        seed = self._seed_int(seed)
        cs = self._covsqrt_hc(eng)
        nc = self.ncomp
        tab = [[cs[i][j] if self._nz[i][j] else None for j in range(nc)] for i in range(nc)]
        r = None
        if rot is not None:
            if rot not in ("inverse", "forward") or nc != 3:
                raise ValueError("draw_hc: rot is 'inverse' or 'forward' and needs three components")
            key = (eng.prec, bool(iau), rot)
            if key not in self._rot_dev:
                m = queb_rotmat(self.geom.lmap(), inverse=(rot == "inverse"), iau=iau)
                self._rot_dev[key] = (eng.fullreal_to_hc(eng.to_real(m[0, 0])), eng.fullreal_to_hc(eng.to_real(m[1, 0])))
            r = self._rot_dev[key]
        if out is None:
            out = torch.empty((nc, eng.ny, eng.kp), dtype=eng.cdt, device=eng.device)
        ins = None if inputs is None else [inputs[i] for i in range(nc)]
        eng.grf_mix(seed, tab, rot=r, inputs=ins, filt=filt, scale=scale, out=[out[i] for i in range(nc)])
        return out

    def get_map_from_rand(self, rand, scalar=False, iau=False, harm=False):
        """Reference arithmetic (maps.py:1579-1587) on a caller-supplied complex
        white-noise array ``rand`` (what pixell.rand_gauss_harm would return):
        covsqrt * rand, unitary inverse C2C, real part."""
        torch = _torch()
        from .engine import precision_of
        eng = _engine(self.shape, precision_of(rand))
        z = eng.to_complex(rand)
        planes, lead = _planes(z)
        nc = self.ncomp
        cs = [[eng.to_real(self.covsqrt[i, j]) for j in range(nc)] for i in range(nc)]
        ks = []
        for i in range(nc):
            acc = None
            for j in range(nc):
                term = eng.cmul_real(planes[j].contiguous(), cs[i][j])
                acc = term if acc is None else acc + term
            ks.append(acc)
        if harm:
            return _ret(torch.stack(ks).reshape(lead + (eng.ny, eng.nx)), rand)
        if not scalar and nc == 3:
            rot = queb_rotmat(self.geom.lmap(), inverse=True, iau=iau)
            ks[1], ks[2] = eng.rot2(eng.to_real(rot[0, 0]), eng.to_real(rot[1, 0]), ks[1], ks[2])
        outs = [torch.real(eng.cfft(k.contiguous(), inverse=True, scale=1.0 / np.sqrt(eng.npix))).contiguous() for k in ks]
        return _ret(torch.stack(outs).reshape(lead + (eng.ny, eng.nx)), rand)


def spec1d_to_2d(shape, wcs, ells, cls):
    """Isotropic 1-D spectrum -> per-mode (Ny,Nx) plane by linear interpolation,
    0 outside the table (SURVEY.md section 8d synthetic-input convention; the
    reference's maps.spec1d_to_2d, maps.py:1590-1591, uses pixell.spec2flat)."""
    ml = as_geometry(shape, wcs).modlmap()
    return np.interp(ml, ells, cls, left=0.0, right=0.0)


def binned_power(imap, bin_edges=None, binner=None, fc=None, modlmap=None, imap2=None, mask=1, wcs=None):
    """maps.py:1350-1361 (``wcs`` is explicit because arrays carry none here)."""
    from . import stats
    shape = imap.shape
    if fc is None:
        fc = FourierCalc(shape, wcs)
    modlmap = fc.geom.modlmap() if modlmap is None else modlmap
    binner = stats.bin2D(modlmap, bin_edges) if binner is None else binner
    p2d, _, _ = fc.power2d(imap * mask, imap2 * mask if imap2 is not None else None)
    cents, p1d = binner.bin(p2d)
    return cents, p1d / np.mean(mask ** 2.)


# ---- Wiener / inverse-variance per-mode filters -----------------------
def matched_filter(imap, fwhm_arcmin, cls=None, noise_uk_arcmin=None, taper_per=12.0, wcs=None):
    """maps.py:677-699: taper -> FFT -> beam/(S+N) (non-finite -> 0) -> inverse FFT, real part.
    ``wcs`` is explicit (arrays carry no geometry here)."""
    from . import cosmology
    geom = as_geometry(imap.shape, wcs)
    taper = 1.0
    if taper_per is not None:
        taper, _ = get_taper(imap.shape[-2:], geom, taper_percent=taper_per)
    modlmap = geom.modlmap()
    p2d = gauss_beam(modlmap, fwhm_arcmin)
    if cls is None:
        s2d = cosmology.default_theory().lCl('TT', modlmap) * p2d ** 2.
    else:
        s2d = np.interp(modlmap, np.arange(cls.size), cls, left=0.0, right=0.0)
    n2d = 0.
    if noise_uk_arcmin is not None:
        n2d = (noise_uk_arcmin * np.pi / 180. / 60.) ** 2.
    with np.errstate(divide="ignore", invalid="ignore"):
        filt2d = p2d / (s2d + n2d)
    filt2d[~np.isfinite(filt2d)] = 0.
    # enmap.fft / enmap.ifft (both unitary) bracket the multiply == filter_map's unnormalised/normalised pair
    return filter_map(imap * taper, filt2d)


def kspace_coadd(kmaps, kbeams, kncovs, fkbeam=1):
    """maps.py:1098-1114: f = sum_i k_i b_i fk / n_i  /  sum_i b_i^2 / n_i per mode (non-finite -> 0).
    The per-mode weights are real one-off planes; the complex accumulation runs on the device."""
    torch = _torch()
    from .engine import precision_of
    kmaps_in = kmaps
    kbeams = np.asarray(kbeams, dtype=np.float64)
    kncovs = np.asarray(kncovs, dtype=np.float64)
    first = kmaps[0]
    eng = _engine(first.shape, precision_of(first))
    with np.errstate(divide="ignore", invalid="ignore"):
        w = kbeams * fkbeam / kncovs
        denom = np.sum(kbeams ** 2 / kncovs, axis=0)
        w[~np.isfinite(w)] = 0            # numer[~finite] = 0 mode by mode
        inv = 1.0 / denom
        inv[~np.isfinite(inv)] = 0
    acc = None
    for i in range(len(kmaps)):
        k = eng.to_complex(kmaps[i])
        wi = eng.to_real(np.broadcast_to(w[i], k.shape))
        term = eng.cmul_real(k, wi)
        if acc is None:
            acc = term
        else:
            ar, tr = torch.view_as_real(acc), torch.view_as_real(term)
            eng.axpby(ar, tr, 1.0, 1.0, out=ar)
    out = eng.cmul_real(acc, eng.to_real(np.broadcast_to(inv, acc.shape)))
    return _ret(out, kmaps_in[0] if not isinstance(kmaps_in, np.ndarray) else kmaps_in)


class MatchedFilter(object):
    """maps.py:2576-2604: amplitude of a template in a map with noise power n2d:
    phi = sum conj(t) k norm mask / n  /  sum |t|^2 norm mask / n  (two global reductions,
    done by the deterministic histogram kernel with a single bin)."""

    def __init__(self, shape, wcs, template=None, noise_power=None):
        self.shape = tuple(shape[-2:])
        self.geom = as_geometry(self.shape, wcs)
        self.fc = FourierCalc(self.shape, self.geom)
        self.normfact = self.geom.area / (np.prod(self.shape)) ** 2
        if noise_power is not None:
            self.n2d = noise_power
        if template is not None:
            self.ktemp = self.fc.fft(np.asarray(template, dtype=np.float64))

    def apply(self, imap=None, kmap=None, template=None, ktemplate=None, noise_power=None, kmask=None):
        torch = _torch()
        from .engine import dev_bin
        if kmap is None:
            kmap = self.fc.fft(np.asarray(imap, dtype=np.float64))
        else:
            assert imap is None
        n2d = self.n2d if noise_power is None else noise_power
        if ktemplate is None:
            ktemp = self.ktemp if template is None else self.fc.fft(np.asarray(template, dtype=np.float64))
        else:
            ktemp = ktemplate
        with np.errstate(divide="ignore", invalid="ignore"):
            in2d = 1. / np.asarray(n2d, dtype=np.float64)
        in2d[~np.isfinite(in2d)] = 0
        w = in2d if kmask is None else in2d * np.real(np.asarray(kmask))
        eng = _engine(self.shape, "f64")
        kt, km = eng.to_complex(ktemp), eng.to_complex(kmap)
        wd = eng.to_real(w)
        ids = torch.zeros(kt.numel(), dtype=torch.int32, device=eng.device)
        s_un, _ = dev_bin(eng.f2power(kt, km, self.normfact).reshape(-1), ids, 1, weights=wd.reshape(-1))
        s_tt, _ = dev_bin(eng.f2power(kt, kt, self.normfact).reshape(-1), ids, 1, weights=wd.reshape(-1))
        phi_un = float(s_un[0].item())
        phi_var = 1. / float(s_tt[0].item())
        return phi_un * phi_var, phi_var


# ---- split-based signal / noise power (maps.py:2296-2411) --------------------
def _hp_val(x):
    return x.t if isinstance(x, HalfPlane) else x


def _hp_like(t, like):
    return HalfPlane(t, like.eng) if isinstance(like, HalfPlane) else t


def split_calc(isplits, jsplits, icoadd, jcoadd, fourier_calc=None, alt=True, wcs=None):
    """maps.py:2296-2333: (total, mean cross-split, noise) 2-D power from the Fourier transforms of n splits of two maps
    and their coadds.  ``isplits`` / ``jsplits``: (nsplits, Ny, Nx) complex arrays or lists of HalfPlane.

    ``alt``: noise = sum_i P(i_i - icoadd, j_i - jcoadd) / ((1 - 1/n) n^2), crosses = total - noise.
    otherwise: crosses = mean over i != j of P(i_i, j_j), noise = total - crosses.  P is bilinear, so the sum over ALL
    ordered pairs is the power of the summed splits: sum_{i != j} P(i_i, j_j) = P(sum_i i_i, sum_j j_j) - sum_i P(i_i, j_i)
    -- n + 1 power evaluations instead of the reference's n (n - 1)."""
    first = icoadd if not isinstance(icoadd, HalfPlane) else None
    fc = fourier_calc if fourier_calc is not None else FourierCalc(tuple(np.shape(first))[-2:] if first is not None else icoadd.shape[-2:], wcs)
    I = [isplits[k] for k in range(len(isplits))]
    J = [jsplits[k] for k in range(len(jsplits))]
    ni, nj = len(I), len(J)
    total = fc.f2power(icoadd, jcoadd)
    power = lambda a, b: _hp_val(fc.f2power(_hp_like(a, icoadd), _hp_like(b, jcoadd)))      # noqa: E731
    if alt:
        assert ni == nj
        spread = sum(power(_hp_val(a) - _hp_val(icoadd), _hp_val(b) - _hp_val(jcoadd)) for a, b in zip(I, J))
        noise = spread / ((1. - 1. / ni) * ni ** 2)
        crosses = _hp_val(total) - noise
    else:
        same = min(ni, nj)
        all_pairs = power(sum(_hp_val(a) for a in I), sum(_hp_val(b) for b in J))
        diagonal = sum(power(_hp_val(I[k]), _hp_val(J[k])) for k in range(same))
        crosses = (all_pairs - diagonal) / float(ni * nj - same)
        noise = _hp_val(total) - crosses
    return total, _hp_like(crosses, total), _hp_like(noise, total)


def noise_from_splits(splits, fourier_calc=None, nthread=0, do_cross=True, wcs=None):
    """maps.py:2338-2411: 2-D noise power of I,Q,U (or T-only) split maps, ``(mean auto - mean cross) / nsplits``, and --
    with ``do_cross`` -- the mean cross power of the splits' T,E,B.  ``splits``: (nsplits, ncomp, Ny, Nx) or
    (nsplits, Ny, Nx) real maps; returns ``(noise, cross_teb)`` as ``power2d`` matrices.

    The reference loops over all n (n - 1) / 2 ordered pairs i < j.  The 2-D power P(a, b) is bilinear, so

        sum_{i < j} P(k_i, k_j) = sum_{j >= 1} P(k_0 + ... + k_{j-1}, k_j):

    one power evaluation per split against the running sum of its predecessors (n - 1 instead of n (n - 1) / 2; the
    component order of every cross term -- first leg from the earlier split -- is the reference's)."""
    maps_in = np.asarray(splits).astype(np.float32)
    if maps_in.ndim not in (3, 4):
        raise AssertionError("splits must be (nsplits, Ny, Nx) or (nsplits, ncomp, Ny, Nx)")
    flat_input = maps_in.ndim == 3
    if flat_input:
        maps_in = maps_in[:, None]
    n, ncomp = maps_in.shape[0], maps_in.shape[1]
    if do_cross and ncomp not in (1, 3):
        raise AssertionError("do_cross needs T-only or I,Q,U splits")
    fc = fourier_calc if fourier_calc is not None else FourierCalc(maps_in.shape[-3:] if do_cross else maps_in.shape[-2:], wcs)

    def transforms(rotate):
        return [fc.iqu2teb(m, nthread=nthread, normalize=False, rot=rotate) for m in maps_in]

    def pair_sum_and_autos(ks, want_auto):
        """(sum_{i<j} P(k_i, k_j), sum_i P(k_i, k_i)) through the running sum of the transforms"""
        run, pairs, autos = None, 0., 0.
        for k in ks:
            if want_auto:
                autos = autos + fc.power2d(kmap=k)[0]
            if run is not None:
                pairs = pairs + fc.power2d(kmap=run, kmap2=k)[0]
                run = run + k
            else:
                run = np.array(k)           # (a copy: the running sum must not alias the first transform)
        return pairs, autos
    npairs = n * (n - 1) / 2.
    pairs, autos = pair_sum_and_autos(transforms(False), True)
    noise = (autos / n - pairs / npairs) / n
    cross_teb = None
    if do_cross:
        # (the reference rotates Q,U -> E,B only when the input was 3-D AND had three components, which cannot both
        # hold once a component axis has been added: maps.py:2376-2378 -- reproduced, its fixtures pin it)
        cross_teb = pair_sum_and_autos(transforms(flat_input and ncomp == 3), False)[0] / npairs
    return noise, cross_teb
