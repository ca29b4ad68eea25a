"""The three host-side theory helpers the hot path needs (SURVEY.md section 2,
cosmology row): a TheorySpectra container with the ``lCl/uCl/gCl`` surface the
reference gets from pyfisher (cosmology.py:863-946), ``power_from_theory``
(cosmology.py:1270-1280), the white-noise level and the Knox bandpower error.
"""
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "cosmo2017_cls.npz")


class TheorySpectra(object):
    """Linear interpolation of tabulated C_ell, zero outside the table and
    beyond ``lpad`` (pyfisher.TheorySpectra behaviour as used by
    loadTheorySpectraFromCAMB, cosmology.py:895-898)."""

    def __init__(self):
        self._u, self._l, self._g = {}, {}, {}
        self.dimensionless = False

    def loadCls(self, ell, Cl, XYType="TT", lensed=False, interporder="linear", lpad=9000, fill_zero=True):
        ell = np.asarray(ell, dtype=np.float64)
        Cl = np.asarray(Cl, dtype=np.float64)
        keep = ell <= lpad if lpad is not None else np.ones(ell.shape, bool)
        (self._l if lensed else self._u)[XYType] = (ell[keep], Cl[keep])

    def loadGenericCls(self, ell, Cl, keyName, lpad=9000, fill_zero=True):
        ell = np.asarray(ell, dtype=np.float64)
        Cl = np.asarray(Cl, dtype=np.float64)
        keep = ell <= lpad if lpad is not None else np.ones(ell.shape, bool)
        self._g[keyName] = (ell[keep], Cl[keep])

    @staticmethod
    def _eval(tab, key, ell):
        if key not in tab and key[::-1] in tab:
            key = key[::-1]
        e, c = tab[key]
        return np.interp(np.asarray(ell, dtype=np.float64), e, c, left=0.0, right=0.0)

    def uCl(self, XYType, ell):
        return self._eval(self._u, XYType, ell)

    def lCl(self, XYType, ell):
        return self._eval(self._l, XYType, ell)

    def gCl(self, keyName, ell):
        return self._eval(self._g, keyName, ell)


def default_theory(lpad=9000):
    """cosmology.default_theory (cosmology.py:850-852): lensed + unlensed
    cosmo2017 spectra in muK^2 and C^kk, from the compact table derived from the
    reference's CAMB output by tests/golden/make_theory_table.py."""
    d = np.load(_DATA)
    th = TheorySpectra()
    for s in ("TT", "EE", "BB", "TE"):
        th.loadCls(d["l_ell"], d["l_" + s], s, lensed=True, lpad=lpad)
    for s in ("TT", "EE", "TE"):
        th.loadCls(d["u_ell"], d["u_" + s], s, lensed=False, lpad=lpad)
    th.loadCls(d["u_ell"], d["u_EE"] * 0., "BB", lensed=False, lpad=lpad)
    th.loadGenericCls(d["kk_ell"], d["kk"], "kk", lpad=lpad)
    return th


def analytic_theory(lmax=20000, A=6.0e3, l0=80., alpha=2.6, ld=1400.):
    """File-free fallback spectra C_l = A (l/l0)^-alpha exp(-(l/ld)^2) (SURVEY.md
    section 8d); EE = 0.05 TT, TE = 0.1 TT with alternating sign suppressed, BB = 0
    unlensed / 1e-4 TT lensed, kk ~ CAMB-like power law.  Shapes only -- for benchmarks."""
    ell = np.arange(2, lmax, dtype=np.float64)
    tt = A * (ell / l0) ** (-alpha) * np.exp(-(ell / ld) ** 2)
    th = TheorySpectra()
    for lensed in (True, False):
        th.loadCls(ell, tt, "TT", lensed=lensed, lpad=None)
        th.loadCls(ell, 0.05 * tt, "EE", lensed=lensed, lpad=None)
        th.loadCls(ell, 0.1 * tt, "TE", lensed=lensed, lpad=None)
        th.loadCls(ell, (1e-4 if lensed else 0.0) * tt, "BB", lensed=lensed, lpad=None)
    kk = 2.5e-7 * (ell / 60.) ** 0.3 / (1 + (ell / 60.) ** 1.6)
    th.loadGenericCls(ell, kk, "kk", lpad=None)
    return th


def power_from_theory(ells, theory, lensed=True, pol=False):
    """cosmology.py:1270-1280."""
    ells = np.asarray(ells)
    ncomp = 3 if pol else 1
    cfunc = theory.lCl if lensed else theory.uCl
    ps = np.zeros((ncomp, ncomp,) + ells.shape)
    ps[0, 0] = cfunc('TT', ells)
    if pol:
        ps[1, 1] = cfunc('EE', ells)
        ps[2, 2] = cfunc('BB', ells)
        ps[0, 1] = cfunc('TE', ells)
        ps[1, 0] = cfunc('TE', ells)
    return ps


def white_noise_power(noise_uk_arcmin):
    """lensing.py:483-488: (sigma * pi/180/60)^2."""
    return (noise_uk_arcmin * np.pi / 180. / 60.) ** 2.


def knox_cov(cl_tot, nmodes):
    """Gaussian bandpower variance 2 C_b^2 / N_modes (mode-count form of
    LensForecast.KnoxCov, cosmology.py:1054-1082, with N_modes from bin2D counts)."""
    return 2. * np.asarray(cl_tot) ** 2. / np.asarray(nmodes)


class LensForecast(object):
    """Gaussian ("Knox") bandpower covariance for lensing / generic spectra -- the output-side contract of
    cosmology.py:948-1094 that the hot path's N_L curves feed: register signal spectra (and optional noise curves)
    under two-letter names like 'kk', then ask for the covariance of band-averaged C^XY with C^WZ.

    Band average: ell-weighted mean over the integers ell_lo..ell_hi of C_ell (+ N_ell for auto-spectra);
    cov(C^XY_b, C^WZ_b) = (C^XW_b C^YZ_b + C^XZ_b C^YW_b) / ((2 ell_mid + 1) (ell_hi - ell_lo) fsky)."""

    def __init__(self, theory=None):
        self.theory = TheorySpectra() if theory is None else theory
        self.Nls = {}

    @staticmethod
    def _noise_curve(ells, nls):
        ells, nls = np.asarray(ells, dtype=float), np.asarray(nls, dtype=float)
        return lambda x: np.interp(np.asarray(x, dtype=float), ells, nls, left=np.inf, right=np.inf)   # unknown noise = no information

    def loadKK(self, ellsCls, Cls, ellsNls, Nls, lpad=30000):
        """kappa-kappa signal + reconstruction noise (cosmology.py:976-985)."""
        self.Nls['kk'] = self._noise_curve(ellsNls, Nls)
        self.theory.loadGenericCls(ellsCls, Cls, 'kk', lpad=lpad)

    def loadGenericCls(self, specType, ellsCls, Cls, ellsNls=None, Nls=None):
        """Any other spectrum, e.g. 'kg', 'gg' (cosmology.py:1034-1036); noise only makes sense for autos."""
        if Nls is not None:
            self.Nls[specType] = self._noise_curve(ellsNls, Nls)
        self.theory.loadGenericCls(ellsCls, Cls, specType)

    def _band_average(self, spec, lo, hi, with_noise=True, noise_only=False):
        ells = np.arange(lo, hi + 1, 1)
        auto = spec[0] == spec[1]
        noise = self.Nls[spec](ells) if (with_noise and auto) else 0.
        total = noise if (noise_only and with_noise and auto) else self.theory.gCl(spec, ells) + noise
        return float(np.sum(ells * total) / np.sum(ells))

    _bin_cls = _band_average        # name used by reference-era callers

    def KnoxCov(self, specTypeXY, specTypeWZ, ellBinEdges, fsky, ntot=False):
        """(variance per band, (S/N)^2 per band of XY, of WZ)  (cosmology.py:1054-1082)."""
        X, Y = specTypeXY
        W, Z = specTypeWZ
        edges = np.asarray(ellBinEdges)
        var, snr2_xy, snr2_wz = [], [], []
        for lo, hi in zip(edges[:-1], edges[1:]):
            b = lambda sp: self._band_average(sp, lo, hi, noise_only=ntot)   # noqa: E731
            pairs = b(X + W) * b(Y + Z) + b(X + Z) * b(Y + W)
            v = pairs / (2. * (0.5 * (lo + hi)) + 1.) / (hi - lo) / fsky
            var.append(v)
            with np.errstate(divide="ignore"):
                inv = np.nan_to_num(1. / v)
            snr2_xy.append(self._band_average(specTypeXY, lo, hi, with_noise=False) ** 2. * inv)
            snr2_wz.append(self._band_average(specTypeWZ, lo, hi, with_noise=False) ** 2. * inv)
        return np.array(var), np.array(snr2_xy), np.array(snr2_wz)

    def sigmaClSquared(self, specType, ellBinEdges, fsky, ntot=False):
        return self.KnoxCov(specType, specType, ellBinEdges, fsky, ntot=ntot)[0]

    def sn(self, ellBinEdges, fsky, specType, ntot=False):
        """Total signal-to-noise and per-band sigma(C_b) (cosmology.py:1087-1094)."""
        var, snr2, _ = self.KnoxCov(specType, specType, ellBinEdges, fsky, ntot=ntot)
        return np.sqrt(snr2.sum()), np.sqrt(var)
