"""NumPy flat-sky lensing quadratic estimator (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED: ``orphics.lensing.Estimator`` / ``qest`` is absent from the
reference snapshot; only its call contract survives
(lensing.py:959-1003, tutorials/tt_verification.ipynb cells 3-4).  This is an
independent float64 / complex128, full-plane C2C implementation of the
Hu & Okamoto (2002) estimator in the real-space form of Hu, DeDeo & Vale (2007),
validated by tests/test_qe_oracle.py (brute-force normalisation on a small
grid, linear response to an injected lens, N0 = A_L for Gaussian fields).

Conventions (DFT = unnormalised forward transform X_l = sum_x X(x) e^{-i l.x},
a = pixel area, continuum X(l) = a X_l):

  <T(l1) T(l2)>_CMB = f(l1,l2) phi(L),  L = l1+l2,
  f_TT = C_l1 (L.l1) + C_l2 (L.l2)                       (HO02 Table 1)
  weight g(l1,l2) = (L.l1) C^g_l1 / (Ct_l1 Ct_l2)         (HDV07 separable form)
  u(L)  = sum over l1 of g T(l1) T(l2) = -i L . FT[ G(x) H(x) ]
          G = IFT[i l Wg T],  H = IFT[Wh T],
          Wg = C^g /(B Ct) mask,  Wh = 1/(B Ct) mask,  Ct = C^tot + N/B^2
  R(L)  = (1/Area) sum_l1 g f   (response; A_L = 1/R is also N0 of phi when the
          filter spectra equal the true ones)
  kappa_hat_l = [L(L+1)/2] u(L) / R(L)         (orphics kappa<->phi, lensing.py:662-665)
"""
import numpy as np

from . import maps_oracle as mo


def _ifftn(a):
    return mo._ifft2(a)


def _fft(a):
    return mo._fft2(a)


class QEOracleTT(object):
    def __init__(self, shape, step_y, step_x, cl_grad2d, cl_tot2d, noise2d, beam2d, kmask, kmask_K=None,
                 grad_cut=None, area=None, cl_resp2d=None):
        """All spectra are 2-D float64 planes on the full (Ny,Nx) lmap grid.
        cl_grad2d: spectrum in the gradient-leg filter; cl_tot2d: signal part of
        the total power (lensed TT); noise2d: noise power (not deconvolved);
        beam2d: beam transfer; kmask: 0/1 T mask; kmask_K: 0/1 kappa mask;
        cl_resp2d: spectrum in the response f (default cl_grad2d)."""
        self.shape = tuple(shape[-2:])
        Ny, Nx = self.shape
        self.area = mo.planar_area(shape, step_y, step_x) if area is None else area
        self.pixarea = self.area / (Ny * Nx)
        self.ly, self.lx = mo.laxes(shape, step_y, step_x)
        self.LY = self.ly[:, None] * np.ones((1, Nx))
        self.LX = np.ones((Ny, 1)) * self.lx[None, :]
        self.modl = np.sqrt(self.LY ** 2 + self.LX ** 2)
        # derivative axes: the self-conjugate Nyquist frequency of a real field has no odd
        # (i*l) component -- exactly what np.real(ifft2(1j*l*F)) discards -- so spectral
        # derivatives use l = 0 there (keeps every returned FT Hermitian).
        lyd, lxd = self.ly.copy(), self.lx.copy()
        lyd[Ny // 2] = 0.0
        lxd[Nx // 2] = 0.0
        self.LYd = lyd[:, None] * np.ones((1, Nx))
        self.LXd = np.ones((Ny, 1)) * lxd[None, :]
        kmask = np.asarray(kmask, dtype=np.float64)
        gmask = kmask.copy()
        if grad_cut is not None:
            gmask[self.modl > grad_cut] = 0
        with np.errstate(divide="ignore", invalid="ignore"):
            ct = cl_tot2d + noise2d / beam2d ** 2
            self.Wg = np.nan_to_num(cl_grad2d / (beam2d * ct), nan=0.0, posinf=0.0, neginf=0.0) * gmask
            self.Wh = np.nan_to_num(1.0 / (beam2d * ct), nan=0.0, posinf=0.0, neginf=0.0) * kmask
            # weights acting on the beam-deconvolved field (used in the response)
            wg = np.nan_to_num(cl_grad2d / ct, nan=0.0, posinf=0.0, neginf=0.0) * gmask
            wh = np.nan_to_num(1.0 / ct, nan=0.0, posinf=0.0, neginf=0.0) * kmask
        cr = cl_grad2d if cl_resp2d is None else cl_resp2d
        self.kmask_K = np.ones(self.shape) if kmask_K is None else np.asarray(kmask_K, dtype=np.float64)
        self.R = self._response(wg, wh, cr)
        with np.errstate(divide="ignore", invalid="ignore"):
            self.AL = np.nan_to_num(1.0 / self.R, nan=0.0, posinf=0.0, neginf=0.0)   # phi normalisation = N0_phi
            self.Fnorm = -(self.modl * (self.modl + 1.) / 2.) * self.AL * self.kmask_K
        self.Nlkk = (self.modl * (self.modl + 1.)) ** 2 / 4. * self.AL

    @classmethod
    def for_timing(cls, shape, step_y, step_x, Wg, Wh, Fnorm):
        """Instance with caller-supplied filter planes and NO response computation
        (bench.py cpu_baseline times only the per-map path)."""
        self = cls.__new__(cls)
        self.shape = tuple(shape[-2:])
        Ny, Nx = self.shape
        ly, lx = mo.laxes(shape, step_y, step_x)
        lyd, lxd = ly.copy(), lx.copy()
        lyd[Ny // 2] = 0.0
        lxd[Nx // 2] = 0.0
        self.LYd = lyd[:, None] * np.ones((1, Nx))
        self.LXd = np.ones((Ny, 1)) * lxd[None, :]
        self.Wg, self.Wh, self.Fnorm = Wg, Wh, Fnorm
        return self

    def _response(self, wg, wh, cr):
        """R(L) = (1/a) sum_jk L_j L_k DFT[ alpha_jk beta + gamma_j delta_k ](L)."""
        l = (self.LXd, self.LYd)
        beta = _ifftn(wh)
        R = np.zeros(self.shape)
        for j in range(2):
            gam = _ifftn(l[j] * wg)
            for k in range(2):
                alpha = _ifftn(l[j] * l[k] * wg * cr)
                delta = _ifftn(l[k] * wh * cr)
                R += (l[j] * l[k] * _fft(alpha * beta + gam * delta)).real
        return R / self.pixarea

    def unnormalized_ft(self, kX, kY):
        """i (lx DFT[gx h] + ly DFT[gy h]) on full-plane DFTs kX (gradient leg), kY."""
        gx = _ifftn(1j * self.LXd * self.Wg * kX).real
        gy = _ifftn(1j * self.LYd * self.Wg * kX).real
        h = _ifftn(self.Wh * kY).real
        return 1j * (self.LXd * _fft(gx * h) + self.LYd * _fft(gy * h))

    def kappa_ft(self, kX, kY=None):
        """DFT of the reconstructed kappa map (same convention as fc.fft(kappa))."""
        if kY is None:
            kY = kX
        return self.Fnorm * self.unnormalized_ft(kX, kY)

    def kappa_from_map(self, XY, T2DData, T2DDataY=None, alreadyFTed=False, returnFt=False):
        """qest.kappa_from_map contract (lensing.py:973-976) for XY == 'TT'."""
        assert XY == "TT"
        kX = np.asarray(T2DData) if alreadyFTed else _fft(np.asarray(T2DData))
        kY = kX if T2DDataY is None else (np.asarray(T2DDataY) if alreadyFTed else _fft(np.asarray(T2DDataY)))
        kft = self.kappa_ft(kX, kY)
        if returnFt:
            return kft
        return _ifftn(kft).real


def brute_force_response_tt(ly, lx, area, wg, wh, cr, Lyi, Lxi):
    """O(N^2) direct sum of R(L) at the single mode (Lyi, Lxi) -- known-answer
    check of the FFT-convolution form on a small grid."""
    Ny, Nx = wg.shape
    LY, LX = ly[Lyi], lx[Lxi]
    tot = 0.0
    for y1 in range(Ny):
        y2 = (Lyi - y1) % Ny
        for x1 in range(Nx):
            x2 = (Lxi - x1) % Nx
            l1y, l1x = ly[y1], lx[x1]
            l2y, l2x = ly[y2], lx[x2]
            Ll1 = LY * l1y + LX * l1x
            Ll2 = LY * l2y + LX * l2x
            g = Ll1 * wg[y1, x1] * wh[y2, x2]
            f = cr[y1, x1] * Ll1 + cr[y2, x2] * Ll2
            tot += g * f
    return tot / area
