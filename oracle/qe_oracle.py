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


# =============================================================================
# General separable estimators (TT, TE, EE, EB, TB) -- same conventions.
#
# E/B are defined by the reference's rotation [E;B] = R(a)[Q;U], a = -2 atan2(-lx, ly)
# (maps.py:1607 / pixell queb_rotmat, non-IAU).  Lensing remaps Q,U as scalars, hence
#   dE(l) = -sum (l1.l2)[cos(D) E(l1) - sin(D) B(l1)] phi(l2),   D = a(l) - a(l1)
#   dB(l) = -sum (l1.l2)[sin(D) E(l1) + cos(D) B(l1)] phi(l2)
# and, with X at l1, Y at l2, D = a2 - a1, L = l1 + l2 (derivation in DESIGN.md section 7):
#   f_TT = C^TT_1 (L.l1) + C^TT_2 (L.l2)
#   f_TE = C^TE_1 cos D (L.l1) + C^TE_2 (L.l2)
#   f_TB = C^TE_1 sin D (L.l1)
#   f_EE = cos D [C^EE_1 (L.l1) + C^EE_2 (L.l2)]
#   f_EB = sin D [C^EE_1 (L.l1) + C^BB_2 (L.l2)]
# Weights (Hu-DeDeo-Vale separable forms; any weight is unbiased once normalised by R = sum g f):
#   g_XY = (L.l1) C^{XY'}_1 trig(D) / (Ct^XX_1 Ct^YY_2)   [+ (L.l2) C^TE_2/(Ct^TT_1 Ct^EE_2) for TE]
# A term is (coef, p, A, B, trig): coef * (L.l_p) * A(l1) * B(l2) * trig(D), trig in {"1","cos","sin"}.
# =============================================================================
def estimator_terms(XY):
    """(weight terms g, response terms f) by NAME of the planes; resolved by the caller.
    'w?XY' = C^XY/Ct (gradient-leg weight), 'i?X' = 1/Ct^XX, 'cXY' = response spectrum, '1' = ones."""
    if XY == "TT":
        return ([(1., 1, "wTT", "iT", "1")], [(1., 1, "cTT", "1", "1"), (1., 2, "1", "cTT", "1")])
    if XY == "EE":
        return ([(1., 1, "wEE", "iE", "cos")], [(1., 1, "cEE", "1", "cos"), (1., 2, "1", "cEE", "cos")])
    if XY == "EB":
        return ([(1., 1, "wEE_B", "iB", "sin")], [(1., 1, "cEE", "1", "sin"), (1., 2, "1", "cBB", "sin")])
    if XY == "TB":
        return ([(1., 1, "wTE_B", "iB", "sin")], [(1., 1, "cTE", "1", "sin")])
    if XY == "TE":
        return ([(1., 1, "wTE", "iE", "cos"), (1., 2, "iT", "wET", "1")],
                [(1., 1, "cTE", "1", "cos"), (1., 2, "1", "cTE", "1")])
    raise ValueError(XY)


def _trig_product(t1, t2):
    """trig(D)*trig'(D) as a list of (coef, harmonic m, kind) with kind in {'1','cos','sin'} of m*D."""
    key = tuple(sorted((t1, t2)))
    if key == ("1", "1"):
        return [(1., 0, "1")]
    if key == ("1", "cos"):
        return [(1., 1, "cos")]
    if key == ("1", "sin"):
        return [(1., 1, "sin")]
    if key == ("cos", "cos"):
        return [(.5, 0, "1"), (.5, 2, "cos")]
    if key == ("sin", "sin"):
        return [(.5, 0, "1"), (-.5, 2, "cos")]
    if key == ("cos", "sin"):
        return [(.5, 2, "sin")]
    raise ValueError(key)


class QEOracle(object):
    """General flat-sky QE (full-plane NumPy).  ``cl``: dict of 2-D lensed spectra
    TT,EE,BB,TE (gradient/response spectra = these: unlensed_equals_lensed);
    ``noise``: dict T,P of 2-D noise powers; masks: dict T,P (0/1)."""

    def __init__(self, shape, step_y, step_x, cl, noise, beam2d, masks, kmask_K=None, area=None, iau=False):
        self.shape = tuple(shape[-2:])
        Ny, Nx = self.shape
        self.area = mo.planar_area(shape, step_y, step_x) if area is None else area
        self.pixarea = self.area / (Ny * Nx)
        ly, lx = mo.laxes(shape, step_y, step_x)
        self.LY = ly[:, None] * np.ones((1, Nx))
        self.LX = np.ones((Ny, 1)) * lx[None, :]
        self.modl = np.sqrt(self.LY ** 2 + self.LX ** 2)
        lyd, lxd = ly.copy(), lx.copy()
        lyd[Ny // 2] = 0.0
        lxd[Nx // 2] = 0.0
        self.Ld = (np.ones((Ny, 1)) * lxd[None, :], lyd[:, None] * np.ones((1, Nx)))  # (x, y) derivative axes
        sgn = 1 if iau else -1
        self.ang = sgn * 2 * np.arctan2(-self.LX, self.LY)
        self.beam = beam2d
        mT, mP = np.asarray(masks["T"], float), np.asarray(masks["P"], float)
        with np.errstate(divide="ignore", invalid="ignore"):
            ct = {"T": cl["TT"] + noise["T"] / beam2d ** 2, "E": cl["EE"] + noise["P"] / beam2d ** 2,
                  "B": cl["BB"] + noise["P"] / beam2d ** 2}
            inv = {k: np.nan_to_num(1.0 / v, nan=0.0, posinf=0.0, neginf=0.0) for k, v in ct.items()}
        m = {"T": mT, "E": mP, "B": mP}
        self.ct, self.cl = ct, cl
        P = {"1": np.ones(self.shape)}
        for X in "TEB":
            P["i" + X] = inv[X] * m[X]
        P["wTT"] = cl["TT"] * inv["T"] * mT
        P["wEE"] = cl["EE"] * inv["E"] * mP
        P["wEE_B"] = P["wEE"]
        P["wTE"] = cl["TE"] * inv["T"] * mT       # gradient on the T leg
        P["wTE_B"] = P["wTE"]
        P["wET"] = cl["TE"] * inv["E"] * mP       # gradient on the E leg
        for k in ("TT", "EE", "BB", "TE"):
            P["c" + k] = cl[k]
        self.P = P
        self.kmask_K = np.ones(self.shape) if kmask_K is None else np.asarray(kmask_K, float)
        self.R, self.AL, self.Fnorm, self.Nlkk = {}, {}, {}, {}

    # ---- normalisation ---------------------------------------------------------------
    def _conv(self, U, V):
        """(1/Area) sum_l1 U(l1) V(L-l1) for full-plane (possibly complex) U, V."""
        return _fft(_ifftn(U) * _ifftn(V)) / self.pixarea

    def _sum_gf(self, gterms, fterms):
        """(1/Area) sum_l1 g f as a real plane, by FFT convolutions."""
        tot = np.zeros(self.shape, complex)
        for (cg, p, Ag, Bg, tg) in gterms:
            for (cf, q, Af, Bf, tf) in fterms:
                A = (self.P[Ag] if isinstance(Ag, str) else Ag) * (self.P[Af] if isinstance(Af, str) else Af)
                B = (self.P[Bg] if isinstance(Bg, str) else Bg) * (self.P[Bf] if isinstance(Bf, str) else Bf)
                for (ct_, mh, kind) in _trig_product(tg, tf):
                    for j in range(2):
                        for k in range(2):
                            U = A * (self.Ld[j] if p == 1 else 1) * (self.Ld[k] if q == 1 else 1)
                            V = B * (self.Ld[j] if p == 2 else 1) * (self.Ld[k] if q == 2 else 1)
                            LL = self.Ld[j] * self.Ld[k]
                            if kind == "1":
                                acc = self._conv(U, V)
                            else:
                                c1, s1 = np.cos(mh * self.ang), np.sin(mh * self.ang)
                                if kind == "cos":     # cos(m(a2-a1)) = c1 c2 + s1 s2
                                    acc = self._conv(U * c1, V * c1) + self._conv(U * s1, V * s1)
                                else:                 # sin(m(a2-a1)) = s2 c1 - c2 s1
                                    acc = self._conv(U * c1, V * s1) - self._conv(U * s1, V * c1)
                            tot += cg * cf * ct_ * LL * acc
        return tot.real

    def setup(self, XY):
        g, f = estimator_terms(XY)
        R = self._sum_gf(g, f)
        with np.errstate(divide="ignore", invalid="ignore"):
            AL = np.nan_to_num(1.0 / R, nan=0.0, posinf=0.0, neginf=0.0)
        self.R[XY], self.AL[XY] = R, AL
        self.Fnorm[XY] = -(self.modl * (self.modl + 1.) / 2.) * AL * self.kmask_K
        # Gaussian noise of the normalised estimator:
        #   N0_phi = A^2 (1/Area) sum g(1,2)[g(1,2) Ct^XX_1 Ct^YY_2 + g(2,1) Ct^XY_1 Ct^XY_2]
        X, Y = XY[0], XY[1]
        cross = {"TT": self.ct["T"], "EE": self.ct["E"], "BB": self.ct["B"], "TE": self.cl["TE"], "ET": self.cl["TE"]}
        cXY = cross.get(X + Y, np.zeros(self.shape))
        t1 = [(1., p, self.ct[X], self.ct[Y], "1") for p in (1,)]
        g_w1 = [(c, p, A, B, t) for (c, p, A, B, t) in g]
        # first piece: g * g * CtXX_1 CtYY_2  == sum_gf(g, g') with g' = g weighted by the total powers
        gp = [(c, p, self._mul(A, self.ct[X]), self._mul(B, self.ct[Y]), t) for (c, p, A, B, t) in g]
        n1 = self._sum_gf(g_w1, gp)
        # second piece: g(1,2) g(2,1) CtXY_1 CtXY_2 ; g(2,1): p -> 3-p, A <-> B, sin -> -sin
        gs = [(c * (-1. if t == "sin" else 1.), 3 - p, self._mul(B, cXY), self._mul(A, cXY), t) for (c, p, A, B, t) in g]
        n2 = self._sum_gf(g_w1, gs)
        self.Nlkk[XY] = (self.modl * (self.modl + 1.)) ** 2 / 4. * AL ** 2 * (n1 + n2)
        return self

    def _mul(self, A, extra):
        return (self.P[A] if isinstance(A, str) else A) * extra

    # ---- reconstruction ------------------------------------------------------------------
    def unnormalized_ft(self, XY, kX, kY):
        g, _ = estimator_terms(XY)
        c, s = np.cos(self.ang), np.sin(self.ang)
        out = 0
        for (cg, p, A, B, trig) in g:
            # p == 1: gradient on the X leg; p == 2: gradient on the Y leg (swap roles)
            kG, kH = (kX, kY) if p == 1 else (kY, kX)
            FG, FH = (self.P[A], self.P[B]) if p == 1 else (self.P[B], self.P[A])
            FG, FH = FG / self.beam, FH / self.beam
            pieces = {"1": [(1., 1., 1.)], "cos": [(1., c, c), (1., s, s)]}
            # sin D with D = a(H-leg... ) careful: D = a2 - a1 where 1 is the X position
            if trig == "sin":
                if p == 1:   # G at l1, H at l2: sin(a2-a1) = sH cG - cH sG
                    pieces["sin"] = [(1., c, s), (-1., s, c)]
                else:        # G at l2, H at l1: sin(a2-a1) = sG cH - cG sH
                    pieces["sin"] = [(1., s, c), (-1., c, s)]
            for (sg, tg, th) in pieces[trig]:
                gx = _ifftn(1j * self.Ld[0] * FG * tg * kG).real
                gy = _ifftn(1j * self.Ld[1] * FG * tg * kG).real
                h = _ifftn(FH * th * kH).real
                out = out + cg * sg * 1j * (self.Ld[0] * _fft(gx * h) + self.Ld[1] * _fft(gy * h))
        return out

    def kappa_ft(self, XY, kX, kY):
        if XY not in self.Fnorm:
            self.setup(XY)
        return self.Fnorm[XY] * self.unnormalized_ft(XY, kX, kY)


    def kappa_mv_ft(self, k, estimators=("TT", "TE", "EE", "EB", "TB")):
        """sum_a w_a kappa_hat^a with w_a = N_a^-1 / sum_b N_b^-1 (diagonal approximation)."""
        for XY in estimators:
            if XY not in self.Nlkk:
                self.setup(XY)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = {XY: np.nan_to_num(1.0 / self.Nlkk[XY], nan=0.0, posinf=0.0, neginf=0.0) for XY in estimators}
            tot = sum(inv.values())
            w = {XY: np.nan_to_num(inv[XY] / tot, nan=0.0, posinf=0.0, neginf=0.0) for XY in estimators}
            self.Nlkk["MV"] = np.nan_to_num(1.0 / tot, nan=0.0, posinf=0.0, neginf=0.0)
        return sum(w[XY] * self.kappa_ft(XY, k[XY[0]], k[XY[1]]) for XY in estimators)


def brute_force_gf(q, gterms, fterms, Lyi, Lxi):
    """Direct O(N^2) evaluation of (1/Area) sum_l1 g f at one L for QEOracle `q`."""
    Ny, Nx = q.shape
    lx, ly = q.Ld[0][0, :], q.Ld[1][:, 0]
    LY, LX = ly[Lyi], lx[Lxi]
    tot = 0.0
    trig = {"1": lambda d: 1.0, "cos": np.cos, "sin": np.sin}

    def val(terms, y1, x1, y2, x2):
        D = q.ang[y2, x2] - q.ang[y1, x1]
        v = 0.0
        for (c, p, A, B, t) in terms:
            a = (q.P[A] if isinstance(A, str) else A)[y1, x1]
            b = (q.P[B] if isinstance(B, str) else B)[y2, x2]
            Ll = (LY * ly[y1] + LX * lx[x1]) if p == 1 else (LY * ly[y2] + LX * lx[x2])
            v += c * Ll * a * b * trig[t](D)
        return v
    for y1 in range(Ny):
        y2 = (Lyi - y1) % Ny
        for x1 in range(Nx):
            x2 = (Lxi - x1) % Nx
            tot += val(gterms, y1, x1, y2, x2) * val(fterms, y1, x1, y2, x2)
    return tot / q.area



# ---- split-based 4-point estimator (lensing.py:980-1003), evaluated exactly as the reference orders it ---------
def split_cross_estimator(qfrag, qpower, splits):
    """``qfrag(a, b)``: kappa FT from X-leg a and Y-leg b; ``qpower(k1, k2)``: 2-D cross power.  1 + 3n + n(n-1) qfrag
    calls, including the ones on the mean split (the product derives those from the pairwise ones by bilinearity)."""
    splits = np.asarray(splits)
    n = splits.shape[0]
    fn = float(n)
    mean = splits.mean(axis=0)
    k_mean = qfrag(mean, mean)
    diag_sum = 0.
    p_single = 0.
    p_pairs = 0.
    for i in range(n):
        sym_i = (qfrag(splits[i], mean) + qfrag(mean, splits[i])) / 2.
        k_ii = qfrag(splits[i], splits[i])
        diag_sum = diag_sum + k_ii
        resid = sym_i - k_ii / fn
        p_single = p_single + qpower(resid, resid)
        for j in range(i + 1, n):
            sym_ij = (qfrag(splits[i], splits[j]) + qfrag(splits[j], splits[i])) / 2.
            p_pairs = p_pairs + qpower(sym_ij, sym_ij)
    k_c = k_mean - diag_sum / fn ** 2.
    return (fn ** 4. * qpower(k_c, k_c) - 4. * fn ** 2. * p_single + 4. * p_pairs) / fn / (fn - 1.) / (fn - 2.) / (fn - 3.)


# ---- flat-sky lensing of simulated maps (lensing.py:395-454, 651-665), signed-coordinate restatement -----
def fkappa_to_fphi(fkappa, modlmap):
    """lensing.py:662-665: phi_l = 2 kappa_l / (l (l + 1)), zero below l = 2 (same operation order as the reference)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        k = np.nan_to_num(2. * fkappa / modlmap / (modlmap + 1.))
    k[modlmap < 2.] = 0.
    return k


def alpha_from_kappa(kappa, step_y, step_x):
    shape = kappa.shape
    ly, lx = mo.laxes(shape, step_y, step_x)
    lyd, lxd = ly.copy(), lx.copy()
    lyd[shape[0] // 2] = 0.0
    lxd[shape[1] // 2] = 0.0
    ml = np.sqrt(ly[:, None] ** 2 + lx[None, :] ** 2)
    fphi = fkappa_to_fphi(_fft(kappa), ml)
    return _ifftn(1j * lyd[:, None] * fphi).real, _ifftn(1j * lxd[None, :] * fphi).real


def flat_taylens(alpha, imap, step_y, step_x, taylor_order=5):
    from math import factorial
    ay, ax = alpha
    Ny, Nx = imap.shape
    ly, lx = mo.laxes(imap.shape, step_y, step_x)
    lyd, lxd = ly.copy(), lx.copy()
    lyd[Ny // 2] = 0.0
    lxd[Nx // 2] = 0.0
    sx = np.rint(ax / step_x).astype(int)
    sy = np.rint(ay / step_y).astype(int)
    dx = ax - sx * step_x
    dy = ay - sy * step_y
    iy, ix = np.mgrid[0:Ny, 0:Nx]
    yy, xx = (iy + sy) % Ny, (ix + sx) % Nx
    out = imap[yy, xx].copy()
    kmap = _fft(imap)
    for n in range(1, taylor_order):
        for b in range(n + 1):
            a = n - b
            d = _ifftn((1j * lxd[None, :]) ** a * (1j * lyd[:, None]) ** b * kmap).real
            out += d[yy, xx] * dx ** a * dy ** b / (factorial(a) * factorial(b))
    return out


def lensed_bb_brute(ly, lx, area, ang, clee, clpp, yi, xi):
    """Direct sum of C^BB(l) = (1/Area) sum_l1 [l1.l2]^2 sin^2(a1 - a) C^EE(l1) C^pp(l2) at one mode."""
    Ny, Nx = clee.shape
    tot = 0.0
    for y1 in range(Ny):
        y2 = (yi - y1) % Ny
        for x1 in range(Nx):
            x2 = (xi - x1) % Nx
            dot = ly[y1] * ly[y2] + lx[x1] * lx[x2]
            tot += dot ** 2 * np.sin(ang[y1, x1] - ang[yi, xi]) ** 2 * clee[y1, x1] * clpp[y2, x2]
    return tot / area
