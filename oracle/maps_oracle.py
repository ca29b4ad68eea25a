"""NumPy restatement of the orphics.maps hot path (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/orphics/maps.py; arithmetic is float64/complex128 with
full-plane C2C FFTs exactly like the reference (maps.py:1613).  The pixell
calls the reference makes are restated from pixell's public behaviour
 -> parity unpinned at that boundary.

Geometry is an explicit value: (Ny, Nx, step_y, step_x, area) with *signed*
steps in radians (standard CAR: step_x < 0).
"""
import numpy as np

try:  # scipy's pocketfft is threaded; numpy's is the fallback
    import scipy.fft as _fft
    _HAVE_SCIPY = True
except Exception:  # pragma: no cover
    import numpy.fft as _fft
    _HAVE_SCIPY = False

_WORKERS = 1


def set_workers(n):
    """FFT threads used by the oracle (bench.py cpu_baseline reports this)."""
    global _WORKERS
    _WORKERS = int(n)


def _fft2(a):
    if _HAVE_SCIPY:
        return _fft.fft2(a, axes=(-2, -1), workers=_WORKERS)
    return _fft.fft2(a, axes=(-2, -1))


def _ifft2(a):
    # numpy/scipy ifft2 divides by Npix == pixell.fft.ifft(normalize=True)
    if _HAVE_SCIPY:
        return _fft.ifft2(a, axes=(-2, -1), workers=_WORKERS)
    return _fft.ifft2(a, axes=(-2, -1))


# ----------------------------------------------------------------------------
# geometry (pixell.enmap.laxes / lmap / modlmap restated; signed steps)
# ----------------------------------------------------------------------------
def laxes(shape, step_y, step_x):
    """pixell enmap.laxes: ly = 2*pi*fftfreq(Ny, step_y), same for x."""
    Ny, Nx = shape[-2:]
    ly = np.fft.fftfreq(Ny, step_y) * 2 * np.pi
    lx = np.fft.fftfreq(Nx, step_x) * 2 * np.pi
    return ly, lx


def lmap(shape, step_y, step_x):
    ly, lx = laxes(shape, step_y, step_x)
    out = np.empty((2,) + tuple(shape[-2:]))
    out[0] = ly[:, None]
    out[1] = lx[None, :]
    return out


def modlmap(shape, step_y, step_x):
    ly, lx = laxes(shape, step_y, step_x)
    return np.sqrt(ly[:, None] ** 2 + lx[None, :] ** 2)


def planar_area(shape, step_y, step_x):
    Ny, Nx = shape[-2:]
    return Ny * Nx * abs(step_y) * abs(step_x)


def queb_rotmat(lmap_, inverse=False, iau=False, spin=2):
    """pixell enmap.queb_rotmat: a = sgn*spin*atan2(-lx, ly)."""
    sgn = 1 if iau else -1
    a = sgn * spin * np.arctan2(-lmap_[1], lmap_[0])
    c, s = np.cos(a), np.sin(a)
    if inverse:
        s = -s
    return np.array([[c, -s], [s, c]])


def map_mul(mat, vec):
    """pixell enmap.map_mul: per-pixel matrix product."""
    if mat.ndim == 2:
        return mat * vec
    if mat.ndim == 3:
        return mat * vec
    return np.einsum("ab...,b...->a...", mat, vec)


# ----------------------------------------------------------------------------
# FourierCalc  (maps.py:1594-1677)
# ----------------------------------------------------------------------------
class FourierCalc(object):
    def __init__(self, shape, step_y, step_x, area=None, iau=False):
        """maps.py:1600-1607."""
        self.shape = tuple(shape)
        self.step_y, self.step_x = step_y, step_x
        if area is None:
            area = planar_area(shape, step_y, step_x)
        self.area = area
        self.normfact = area / np.prod(self.shape[-2:]) ** 2.
        if len(shape) > 2 and shape[-3] > 1:
            self.rot = queb_rotmat(lmap(shape, step_y, step_x), iau=iau)

    def iqu2teb(self, emap, nthread=0, normalize=True, rot=True):
        """maps.py:1609-1617 (enmap.fft normalize=True is 1/sqrt(Npix))."""
        k = _fft2(np.asarray(emap))
        if normalize:
            k = k / np.prod(k.shape[-2:]) ** 0.5
        if k.ndim > 2 and k.shape[-3] > 1 and rot:
            k[..., -2:, :, :] = map_mul(self.rot, k[..., -2:, :, :])
        return k

    def f2power(self, kmap1, kmap2, pixel_units=False):
        """maps.py:1620-1624."""
        norm = 1. if pixel_units else self.normfact
        return np.real(np.conjugate(kmap1) * kmap2) * norm

    def f1power(self, map1, kmap2, pixel_units=False, nthread=0):
        """maps.py:1626-1630."""
        kmap1 = self.iqu2teb(map1, nthread, normalize=False)
        norm = 1. if pixel_units else self.normfact
        return np.real(np.conjugate(kmap1) * kmap2) * norm, kmap1

    def ifft(self, kmap):
        """maps.py:1632-1633 (normalize=True -> /Npix)."""
        return _ifft2(np.asarray(kmap))

    def fft(self, emap):
        """maps.py:1635-1636 (unnormalised forward)."""
        return _fft2(np.asarray(emap))

    def power2d(self, emap=None, emap2=None, nthread=0, pixel_units=False,
                skip_cross=False, rot=True, kmap=None, kmap2=None, dtype=None):
        """maps.py:1639-1677."""
        if kmap is not None:
            lteb1 = kmap
            ndim = kmap.ndim
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
            lteb2 = self.iqu2teb(emap2, nthread, normalize=False, rot=rot) \
                if emap2 is not None else lteb1
        assert lteb1.shape == lteb2.shape
        if ndim > 2 and ncomp > 1:
            retpow = np.zeros((ncomp, ncomp, lteb1.shape[-2], lteb1.shape[-1]), dtype=dtype)
            for i in range(ncomp):
                retpow[i, i] = self.f2power(lteb1[i], lteb2[i], pixel_units)
            if not skip_cross:
                for i in range(ncomp):
                    for j in range(i + 1, ncomp):
                        retpow[i, j] = self.f2power(lteb1[i], lteb2[j], pixel_units)
                        retpow[j, i] = retpow[i, j]
            return retpow, lteb1, lteb2
        if lteb1.ndim > 2:
            lteb1 = lteb1[0]
        if lteb2.ndim > 2:
            lteb2 = lteb2[0]
        return self.f2power(lteb1, lteb2, pixel_units), lteb1, lteb2


# ----------------------------------------------------------------------------
# k-space filters (maps.py:1922-1948, 677-699)
# ----------------------------------------------------------------------------
def filter_map(imap, kfilter):
    """maps.py:1922-1923."""
    return np.real(_ifft2(_fft2(np.asarray(imap)) * kfilter))


def gauss_beam(ell, fwhm):
    """maps.py:1925-1927."""
    tht_fwhm = np.deg2rad(fwhm / 60.)
    return np.exp(-(tht_fwhm ** 2.) * (ell ** 2.) / (16. * np.log(2.)))


def mask_kspace(shape, step_y, step_x, lxcut=None, lycut=None, lmin=None, lmax=None):
    """maps.py:1936-1948 (int ones; strict comparisons as in the reference)."""
    output = np.ones(shape[-2:], dtype=int)
    if (lmin is not None) or (lmax is not None):
        ml = modlmap(shape, step_y, step_x)
    if (lxcut is not None) or (lycut is not None):
        ly, lx = laxes(shape, step_y, step_x)
    if lmin is not None:
        output[np.where(ml <= lmin)] = 0
    if lmax is not None:
        output[np.where(ml >= lmax)] = 0
    if lxcut is not None:
        output[:, np.where(np.abs(lx) < lxcut)] = 0
    if lycut is not None:
        output[np.where(np.abs(ly) < lycut), :] = 0
    return output


def matched_filter_weights(beam2d, total_power2d):
    """maps.py:677-699 core: filt2d = beam/(S+N), non-finite -> 0."""
    with np.errstate(divide="ignore", invalid="ignore"):
        filt = beam2d / total_power2d
    filt[~np.isfinite(filt)] = 0
    return filt


def cosine_window(Ny, Nx, lenApodY=30, lenApodX=30, padY=0, padX=0):
    """maps.py:1891-1920."""
    win = np.ones((Ny, Nx))
    i = np.arange(Nx)
    j = np.arange(Ny)
    ii, jj = np.meshgrid(i, j)
    if lenApodX > 0:
        r = ii.astype(float) - padX
        sel = np.where(ii <= (lenApodX + padX))
        win[sel] = 1. / 2 * (1 - np.cos(-np.pi * r[sel] / lenApodX))
        sel = np.where(ii >= ((Nx - 1) - lenApodX - padX))
        r = ((Nx - 1) - ii - padX).astype(float)
        win[sel] = 1. / 2 * (1 - np.cos(-np.pi * r[sel] / lenApodX))
    if lenApodY > 0:
        r = jj.astype(float) - padY
        sel = np.where(jj <= (lenApodY + padY))
        win[sel] *= 1. / 2 * (1 - np.cos(-np.pi * r[sel] / lenApodY))
        sel = np.where(jj >= ((Ny - 1) - lenApodY - padY))
        r = ((Ny - 1) - jj - padY).astype(float)
        win[sel] *= 1. / 2 * (1 - np.cos(-np.pi * r[sel] / lenApodY))
    win[0:padY, :] = 0
    win[:, 0:padX] = 0
    win[Ny - padY:, :] = 0
    win[:, Nx - padX:] = 0
    return win


def get_taper(shape, taper_percent=12.0, pad_percent=3.0, weight=None):
    """maps.py:1873-1878."""
    Ny, Nx = shape[-2:]
    if weight is None:
        weight = np.ones(shape[-2:])
    n = min(Ny, Nx)
    taper = cosine_window(Ny, Nx, lenApodY=int(taper_percent * n / 100.),
                          lenApodX=int(taper_percent * n / 100.),
                          padY=int(pad_percent * n / 100.),
                          padX=int(pad_percent * n / 100.)) * weight
    return taper, np.mean(taper ** 2.)


# ----------------------------------------------------------------------------
# MapGen (maps.py:1553-1587), 4-D covariance path
# ----------------------------------------------------------------------------
def multi_pow(cov, exp):
    """pixell enmap.multi_pow: per-mode symmetric matrix power (eigpow)."""
    nc = cov.shape[0]
    if nc == 1:
        with np.errstate(invalid="ignore"):
            out = np.abs(cov) ** exp
        return out
    m = np.moveaxis(cov.reshape(nc, nc, -1), -1, 0)  # (npix,nc,nc)
    w, v = np.linalg.eigh(m)
    w = np.where(w > 0, w, 0.0) ** exp
    res = np.einsum("pab,pb,pcb->pac", v, w, v)
    return np.moveaxis(res, 0, -1).reshape(cov.shape)


def spec2flat(shape, step_y, step_x, cov, exp=1.0, smooth_width=0.0, area=None):
    """enmap.spec2flat as MapGen uses it (maps.py:1573) -- PARITY UNPINNED (pixell absent): 1-D spectra sampled at
    integer ell -> optional Gaussian smoothing in ell with weight ell -> x Npix/area -> per-ell matrix power ->
    linear interpolation onto |ell|, 0 beyond the table.  Written independently of the product (explicit loops
    over ell for the smoothing and the matrix power) so the two can be compared."""
    cov = np.array(cov, dtype=np.float64)
    if cov.ndim == 1:
        cov = cov[None, None]
    nc, nl = cov.shape[0], cov.shape[-1]
    Ny, Nx = shape[-2:]
    area = planar_area(shape, step_y, step_x) if area is None else area
    if smooth_width > 0:
        ell = np.arange(nl, dtype=np.float64)
        wgt = np.maximum(ell, 0.5)
        sm = np.empty_like(cov)
        half = int(min(nl - 1, np.ceil(5 * smooth_width)))
        for l in range(nl):
            lo, hi = max(0, l - half), min(nl, l + half + 1)
            k = np.exp(-0.5 * ((ell[lo:hi] - l) / smooth_width) ** 2)
            # np.convolve(mode="same") zero-pads: the weight normalisation sees the same truncated window
            sm[..., l] = (cov[..., lo:hi] * (wgt[lo:hi] * k)).sum(-1) / (wgt[lo:hi] * k).sum()
        cov = sm
    cov = cov * (Ny * Nx / area)
    if exp != 1.0:
        out = np.empty_like(cov)
        for l in range(nl):
            w, v = np.linalg.eigh(cov[:, :, l])
            out[:, :, l] = (v * np.where(w > 0, w, 0.0) ** exp) @ v.T
        cov = out
    cov[~np.isfinite(cov)] = 0
    ml = modlmap(shape, step_y, step_x)
    ell = np.arange(nl, dtype=np.float64)
    res = np.zeros((nc, nc) + ml.shape)
    for i in range(nc):
        for j in range(nc):
            res[i, j] = np.interp(ml, ell, cov[i, j], right=0.0)
    return res


class MapGen(object):
    def __init__(self, shape, step_y, step_x, cov=None, covsqrt=None, pixel_units=False, area=None):
        """maps.py:1559-1573 (cov.ndim==4 branch only; the 3-D branch is
        pixell.spec2flat's interpolation and is out of the pinned scope)."""
        self.shape = tuple(shape)
        self.step_y, self.step_x = step_y, step_x
        if area is None:
            area = planar_area(shape, step_y, step_x)
        if covsqrt is not None:
            self.covsqrt = covsqrt
        else:
            assert cov.ndim == 4
            if not pixel_units:
                cov = cov * np.prod(self.shape[-2:]) / area
            self.covsqrt = multi_pow(cov, 0.5)

    def get_map_from_rand(self, rand, scalar=False, iau=False, harm=False):
        """maps.py:1578-1587 with the random draw factored out: ``rand`` is the
        complex array ``randn + 1j*randn`` pixell.rand_gauss_harm would return."""
        if self.covsqrt.shape[0] == 1:
            data = self.covsqrt[0, 0] * rand
            if rand.ndim == 2:
                data = data.reshape(rand.shape)
        else:
            data = np.einsum("ab...,b...->a...", self.covsqrt, rand)
        if harm:
            return data
        npix = np.prod(self.shape[-2:])
        if scalar:
            return (_ifft2(data) * npix ** 0.5).real  # unitary inverse
        # harm2map: rotate E,B -> Q,U (inverse rotation), unitary ifft, real
        data = np.array(data)
        if data.ndim > 2 and data.shape[-3] > 1:
            rot = queb_rotmat(lmap(self.shape, self.step_y, self.step_x), inverse=True, iau=iau)
            data[..., -2:, :, :] = map_mul(rot, data[..., -2:, :, :])
        return (_ifft2(data) * npix ** 0.5).real

    def get_map(self, seed=None, scalar=False, iau=False, real=False, harm=False):
        """maps.py:1576-1587 including the legacy global-RNG draw order."""
        if seed is not None:
            np.random.seed(seed)
        if real:
            rand = _fft2(np.random.standard_normal(self.shape)) / np.prod(self.shape[-2:]) ** 0.5
        else:
            rand = np.random.standard_normal(self.shape) + 1j * np.random.standard_normal(self.shape)
        return self.get_map_from_rand(rand, scalar=scalar, iau=iau, harm=harm)


def white_noise_power(noise_uk_arcmin):
    """lensing.py:483-488: (sigma*pi/180/60)^2."""
    return (noise_uk_arcmin * np.pi / 180. / 60.) ** 2.


def interp_spectrum(ells, cls, ell2d):
    """Linear interpolation of a 1-D C_ell onto a 2-D |ell| grid, 0 outside
    (SURVEY.md section 8d synthetic-input convention)."""
    return np.interp(ell2d, ells, cls, left=0.0, right=0.0)


def matched_filter(imap, fwhm_arcmin, step_y, step_x, cls, noise_uk_arcmin=None, taper_per=12.0):
    """maps.py:677-699 (unitary fft/ifft pair == fft2/ifft2)."""
    shape = imap.shape
    taper = 1.0
    if taper_per is not None:
        taper, _ = get_taper(shape[-2:], taper_percent=taper_per)
    ml = modlmap(shape, step_y, step_x)
    p2d = gauss_beam(ml, fwhm_arcmin)
    s2d = np.interp(ml, np.arange(cls.size), cls, left=0.0, right=0.0)
    n2d = 0. if noise_uk_arcmin is None else (noise_uk_arcmin * np.pi / 180. / 60.) ** 2.
    with np.errstate(divide="ignore", invalid="ignore"):
        filt2d = p2d / (s2d + n2d)
    filt2d[~np.isfinite(filt2d)] = 0.
    return _ifft2(_fft2(imap * taper) * filt2d).real


def kspace_coadd(kmaps, kbeams, kncovs, fkbeam=1):
    """maps.py:1098-1114."""
    kmaps, kbeams, kncovs = np.asarray(kmaps), np.asarray(kbeams), np.asarray(kncovs)
    with np.errstate(divide="ignore", invalid="ignore"):
        numer = np.sum(kmaps * kbeams * fkbeam / kncovs, axis=0)
        numer[~np.isfinite(numer)] = 0
        denom = np.sum(kbeams ** 2 / kncovs, axis=0)
        f = numer / denom
    f[~np.isfinite(f)] = 0
    return f


def split_calc(isplits, jsplits, icoadd, jcoadd, f2power, alt=True):
    """maps.py:2296-2333: (total, mean cross, noise) 2-D power from split transforms; ``f2power(k1, k2)`` is the power
    function (FourierCalc.f2power)."""
    isplits, jsplits = np.asarray(isplits), np.asarray(jsplits)
    total = f2power(icoadd, jcoadd)
    ni, nj = isplits.shape[0], jsplits.shape[0]
    if alt:
        assert ni == nj
        noise = 0.
        for a, b in zip(isplits, jsplits):
            noise = noise + f2power(a - icoadd, b - jcoadd)
        noise = noise / ((1. - 1. / ni) * ni ** 2)
        return total, total - noise, noise
    acc, count = 0., 0.
    for i in range(ni):
        for j in range(nj):
            if i != j:
                acc = acc + f2power(isplits[i], jsplits[j])
                count += 1.
    crosses = acc / count
    return total, crosses, total - crosses


def matched_filter_apply(ktemp, kmap, n2d, normfact, kmask=None):
    """MatchedFilter.apply, maps.py:2587-2604."""
    if kmask is None:
        kmask = 1.0
    with np.errstate(divide="ignore", invalid="ignore"):
        in2d = 1. / n2d
    in2d[~np.isfinite(in2d)] = 0
    phi_un = np.sum(ktemp.conj() * kmap * normfact * kmask * in2d).real
    phi_var = 1. / np.sum(ktemp.conj() * ktemp * normfact * kmask * in2d).real
    return phi_un * phi_var, phi_var
