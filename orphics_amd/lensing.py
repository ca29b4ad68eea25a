"""Flat-sky lensing quadratic estimator + sim harness on MI355X.

``Estimator`` / ``qest`` reproduce the call contract that survives in the
reference (``qest.kappa_from_map``, lensing.py:973-976; constructor keywords
from tutorials/tt_verification.ipynb cell 3).  The class itself is absent from
the reference snapshot: the estimator is the Hu & Okamoto (2002)
one in the real-space form of Hu, DeDeo & Vale (2007); oracle/qe_oracle.py is
the float64 NumPy statement it is tested against.

Per reconstruction (TT): 1 fused leg-filter kernel, 3 C2R FFTs, 2 real
products, 2 R2C FFTs, 1 divergence/normalisation kernel -- all HIP, all on the
half-plane layout.
"""
import numpy as np

from . import maps
from .cosmology import power_from_theory
from .geometry import as_geometry
from .stats import HalfPlane


def _torch():
    import torch
    return torch


def _half(a, nxh):
    """Full-plane (Ny,Nx) host array -> the non-redundant half (Ny,Nx/2+1)."""
    a = np.asarray(a)
    return np.ascontiguousarray(a[:, :nxh + 1])


def _safe_div(num, den):
    with np.errstate(divide="ignore", invalid="ignore"):
        out = num / den
    out[~np.isfinite(out)] = 0
    return out


class Estimator(object):
    """Flat-sky QE.  Keywords follow the reference's ``lensing.qest`` call
    (tt_verification.ipynb cell 3):

    theory  : object with lCl/uCl(spec, ell) (cosmology.TheorySpectra)
    noise2d : (Ny,Nx) T noise power (not beam-deconvolved), beam2d the beam
              transfer, kmask the 0/1 T mask; *_P the polarisation versions;
    kmask_K : 0/1 mask applied to the reconstructed kappa modes
    grad_cut: zero the gradient leg above this ell
    unlensed_equals_lensed: use the lensed spectra in the gradient leg/response
    dtype   : "f32" (fast path) or "f64" (parity mode) for the per-map kernels;
              the one-off normalisation is always computed with the f64 kernels.
    """

    def __init__(self, shape, wcs, theory, noise2d=None, beam2d=None, kmask=None, noise2d_P=None, kmask_P=None,
                 kmask_K=None, pol=False, grad_cut=None, unlensed_equals_lensed=False, bigell=9000, dtype="f32",
                 theory_norm=None, prune=True, iau=False, row_grid="auto", col_grid="auto"):
        # prune: let the fused kernels skip the hc columns on which the (band-limited) filters vanish --
        # same arithmetic on the remaining columns, identical results (include/orphics_amd.h, ACTIVE COLUMNS)
        self.prune = bool(prune)
        self._token = object()        # identity of this handle for the plan-level filter / bin binding
        # row_grid: grid of the fused row stage's real-space products.  "auto" (with prune): the smallest power of two
        # >= 2 leg_cols + kappa_cols -- exact for band-limited filters (include/orphics_amd.h, ROW GRID); None / "full":
        # always the map's own nx points.
        self.mrow = -1 if (self.prune and row_grid == "auto") else (0 if row_grid in (None, "full", "auto") else int(row_grid))
        # col_grid: the same for the y axis of the one-call TT path (include/orphics_amd.h, COLUMN GRID): "auto" = the
        # smallest power of two >= max(2 leg_rows + kappa_rows, 2 kappa_rows) when that is < ny; "full" = the map's ny rows
        self.mcol = -1 if (self.prune and col_grid == "auto" and self.mrow != 0) else (0 if col_grid in (None, "full", "auto") else int(col_grid))
        self.iau = bool(iau)          # polarisation angle convention of the E/B inputs (FourierCalc(iau=...), maps.py:1600)
        self.shape = tuple(shape)
        self.wcs = wcs
        self.geom = as_geometry(shape, wcs)
        self.prec = dtype
        self.pol = pol
        self.eng = maps._engine(self.shape, dtype)
        self.eng64 = maps._engine(self.shape, "f64")
        Ny, Nx = self.shape[-2:]
        self.nxh = Nx // 2
        ly, lx = self.geom.laxes()
        self.eng.set_laxes(ly, lx)
        self.eng64.set_laxes(ly, lx)
        self.ly, self.lxh = ly, lx[:self.nxh + 1]
        self.modl_h = np.sqrt(ly[:, None] ** 2 + self.lxh[None, :] ** 2)
        ones = np.ones((Ny, self.nxh + 1))
        ml = np.where(self.modl_h <= bigell, self.modl_h, -1.0)  # theory is 0 beyond bigell
        cfun_g = theory.lCl if unlensed_equals_lensed else theory.uCl
        self.noise = {"T": _half(noise2d, self.nxh) if noise2d is not None else 0 * ones}
        self.beam = _half(beam2d, self.nxh) if beam2d is not None else ones
        self.mask = {"T": _half(kmask, self.nxh).astype(np.float64) if kmask is not None else ones}
        self.mask_K = _half(kmask_K, self.nxh).astype(np.float64) if kmask_K is not None else ones
        self.grad_cut = grad_cut
        self.cl_grad = {"TT": np.where(ml >= 0, cfun_g("TT", np.abs(ml)), 0.0)}
        self.cl_len = {"TT": np.where(ml >= 0, theory.lCl("TT", np.abs(ml)), 0.0)}
        self.AL, self.Nlkk, self._F, self._W, self._R = {}, {}, {}, {}, {}
        self._work = None
        self._setup_tt()
        if pol:
            for sp in ("EE", "BB", "TE"):
                self.cl_grad[sp] = np.where(ml >= 0, cfun_g(sp, np.abs(ml)), 0.0)
                self.cl_len[sp] = np.where(ml >= 0, theory.lCl(sp, np.abs(ml)), 0.0)
            self.noise["P"] = _half(noise2d_P, self.nxh) if noise2d_P is not None else 2.0 * self.noise["T"]
            self.mask["P"] = _half(kmask_P, self.nxh).astype(np.float64) if kmask_P is not None else self.mask["T"]
            sgn = 1 if self.iau else -1
            lyg, lxg = np.meshgrid(self.ly, self.lxh, indexing="ij")
            self.ang_h = sgn * 2 * np.arctan2(-lxg, lyg)       # pixell queb_rotmat angle (maps.py:1607)
            self._gen = {}

    # ---- filters / normalisation --------------------------------------------------
    def _hcreal(self, eng, a_half):
        """(Ny, Nx/2+1) host plane -> padded hc-layout device plane."""
        torch = _torch()
        t = eng.hcreal()
        t[:, :self.nxh + 1] = torch.as_tensor(np.ascontiguousarray(a_half), dtype=eng.rdt, device=eng.device)
        return t

    def _setup_tt(self):
        gmask = self.mask["T"].copy()
        if self.grad_cut is not None:
            gmask[self.modl_h > self.grad_cut] = 0
        ct = self.cl_len["TT"] + _safe_div(self.noise["T"], self.beam ** 2)
        wg = _safe_div(self.cl_grad["TT"], ct) * gmask          # on the deconvolved field
        wh = _safe_div(np.ones_like(ct), ct) * self.mask["T"]
        Wg = _safe_div(wg, self.beam)                            # on the observed (beam-convolved) field
        Wh = _safe_div(wh, self.beam)
        R = self._response_tt(wg, wh, self.cl_grad["TT"])
        AL = _safe_div(np.ones_like(R), R)
        L = self.modl_h
        Fnorm = -(L * (L + 1.) / 2.) * AL * self.mask_K
        self.AL["TT"] = AL
        self.Nlkk["TT"] = (L * (L + 1.)) ** 2 / 4. * AL
        self.R_TT = R
        self._F["TT"] = (self._hcreal(self.eng, Wg), self._hcreal(self.eng, Wh), self._hcreal(self.eng, Fnorm))
        self._W["TT"] = (self._support_cols(Wg, Wh), self._support_cols(Fnorm))
        self._R["TT"] = (self._support_rows(Wg, Wh), self._support_rows(Fnorm))

    def _support_cols(self, *planes):
        """Number of leading hc columns outside which all the given half-plane filters vanish (0 = no pruning)."""
        if not self.prune:
            return 0
        nz = np.zeros(self.nxh + 1, dtype=bool)
        for a in planes:
            nz |= np.any(np.asarray(a) != 0, axis=0)
        w = int(np.nonzero(nz)[0].max()) + 1 if nz.any() else 1
        return 0 if w >= self.nxh + 1 else w

    def _support_rows(self, *planes):
        """Row band outside which all the given half-plane filters vanish: rows y < rb or y > Ny - rb may be
        non-zero (rb = 1 + the largest |ky index|); 0 = no pruning."""
        if not self.prune:
            return 0
        Ny = self.shape[-2]
        nz = np.zeros(Ny, dtype=bool)
        for a in planes:
            nz |= np.any(np.asarray(a) != 0, axis=1)
        if not nz.any():
            return 1
        y = np.nonzero(nz)[0]
        rb = int(np.minimum(y, Ny - y).max()) + 1
        return 0 if 2 * rb - 1 >= Ny else rb

    @property
    def col_grid(self):
        """Rows the one-call TT path runs its inverse-column / row / forward-column stages on (0 = the map's ny)."""
        return int(self._bind().lib.oa_plan_col_grid(self.eng.plan))

    @property
    def kappa_rows(self):
        """Row band of kappa_hat (``Engine.bin_power(..., active_rows=q.kappa_rows)``)."""
        return self._R["TT"][1]

    def new_output(self):
        """A zero-initialised kappa_hat plane OWNED by the estimator family: pass it as ``out=`` to the
        ``reconstruct_*`` calls of a Monte-Carlo loop.  The pruned kernels write only the active region of kappa_hat;
        for an owned plane the zero-fill of the rest is done once and remembered (``engine._OWNED``), and forgotten
        again as soon as anything else writes into the plane through an Engine wrapper or an in-place torch op.
        Any other tensor passed as ``out=`` is zero-filled outside the active region on EVERY call."""
        from .engine import register_owned
        return register_owned(self.eng.hc())

    @property
    def leg_rows(self):
        """Row band of the input transform the TT estimator reads (``Engine.rfft(..., rband=q.leg_rows)``)."""
        return self._R["TT"][0]

    @property
    def leg_cols(self):
        """Columns of the input transform the TT estimator reads (``Engine.rfft(..., width=q.leg_cols)``)."""
        return self._W["TT"][0]

    @property
    def kappa_cols(self):
        """Columns of kappa_hat that can be non-zero (``Engine.bin_power(..., active_cols=q.kappa_cols)``)."""
        return self._W["TT"][1]

    def _response_tt(self, wg, wh, cr):
        """R(L) = (1/a) sum_jk L_j L_k DFT[alpha_jk beta - gt_j dt_k](L) evaluated with
        the f64 kernels (one-off): alpha_jk = IDFT[l_j l_k wg C], beta = IDFT[wh],
        gt_j = IDFT[i l_j wg], dt_k = IDFT[i l_k wh C]  (oracle/qe_oracle.py)."""
        torch = _torch()
        e = self.eng64
        one_k = e.hc()
        one_k[:, :self.nxh + 1] = 1.0
        F_wg, F_wh = self._hcreal(e, wg), self._hcreal(e, wh)
        F_wgc, F_whc = self._hcreal(e, wg * cr), self._hcreal(e, wh * cr)
        F_one = self._hcreal(e, np.ones_like(wg))
        # first-derivative legs
        kgx, kgy, kb = e.qe_legs(one_k, one_k, F_wg, F_wh)          # i lx wg, i ly wg, wh
        kdx, kdy, _ = e.qe_legs(one_k, one_k, F_whc, F_wh)          # i lx wh C, i ly wh C
        kax, kay, _ = e.qe_legs(one_k, one_k, F_wgc, F_wh)          # i lx wg C, i ly wg C
        kaxx, kaxy, _ = e.qe_legs(kax, one_k, F_one, F_wh)          # -lx lx wgC, -lx ly wgC
        _, kayy, _ = e.qe_legs(kay, one_k, F_one, F_wh)             # -ly ly wgC
        beta = e.irfft(kb)
        gx, gy = e.irfft(kgx), e.irfft(kgy)
        dx, dy = e.irfft(kdx), e.irfft(kdy)
        axx, axy, ayy = e.irfft(kaxx), e.irfft(kaxy), e.irfft(kayy)   # = -alpha_jk
        # S_jk = alpha_jk beta - gt_j dt_k  ->  -S_jk = axx*beta + gx*dx
        sxx = e.axpby(e.mul_real(axx, beta), e.mul_real(gx, dx), 1.0, 1.0)
        syy = e.axpby(e.mul_real(ayy, beta), e.mul_real(gy, dy), 1.0, 1.0)
        sxy = e.axpby(e.mul_real(axy, beta), e.axpby(e.mul_real(gx, dy), e.mul_real(gy, dx), 0.5, 0.5), 1.0, 1.0)
        A, B, C2 = e.rfft(sxx), e.rfft(syy), e.rfft(sxy)             # = -DFT[S_xx], -DFT[S_yy], -DFT[S_xy]/2... (sxy is the half cross term)
        # lx^2 A' + ly^2 B' + 2 lx ly C'  with two divergence passes: i l.(i l.M) = -l.M.l
        ux = e.qe_div(A, C2, F_one)
        uy = e.qe_div(C2, B, F_one)
        tot = e.qe_div(ux, uy, F_one)                                # = -(lx^2 A + 2 lx ly C2 + ly^2 B)
        Rk = e.f2power(tot, one_k, 1.0 / self.geom.pixarea)          # real part / a ; signs: (-1)*(-1) = +
        return Rk.cpu().numpy()[:, :self.nxh + 1].astype(np.float64)

    def fork(self):
        """A second handle on the same estimator for use on ANOTHER HIP stream: shares the (read-only)
        filter / normalisation planes, owns its own C-ABI plan (FFT scratch) and work buffers, so independent
        realisations can be in flight concurrently (Monte-Carlo loops)."""
        import copy
        from .engine import Engine
        other = copy.copy(self)
        other._token = object()
        other.eng = Engine(self.eng.ny, self.eng.nx, self.prec)
        other.eng.set_laxes(*self.geom.laxes())
        self.eng.copy_options_to(other.eng)      # plan options (Engine.set_option) follow the estimator onto its lanes' plans
        other._work = None
        other._rwork = None
        other._acc = None
        return other

    def astype(self, dtype):
        """The same estimator (filters, normalisations, MV weights: all computed once with the f64 kernels and kept on
        the host) with its per-map kernels in another precision: device planes are converted, nothing is recomputed.
        ``q64.astype("f32")`` is how a parity run and a production run share one expensive set-up."""
        import copy
        if dtype == self.prec:
            return self
        other = copy.copy(self)
        other._token = object()
        other.prec = dtype
        other.eng = maps._engine(self.shape, dtype)
        other.eng.set_laxes(*self.geom.laxes())
        rdt = other.eng.rdt
        other._F = {k: tuple(t.to(rdt) for t in v) for k, v in self._F.items()}
        # the tag -> device filter plane cache is PER PRECISION: converted once, and the converted pieces are looked up in
        # it by plane identity so that estimators sharing a filtered field keep sharing ONE plane object (oa_qe_mv
        # deduplicates leg transforms by pointer); estimators set up later on either handle fill their own cache
        fdev = getattr(self, "_fdev", None)
        conv = {}
        if fdev is not None:
            other._fdev = {}
            for tag, t in fdev.items():
                other._fdev[tag] = conv[id(t)] = t.to(rdt)

        def plane(t):
            if id(t) not in conv:
                conv[id(t)] = t.to(rdt)
            return conv[id(t)]
        if getattr(self, "_gen", None) is not None:
            other._gen = {}
            for XY, G in self._gen.items():
                G2 = {k: v for k, v in G.items() if k != "c_args"}
                G2["pieces"] = [(sg, plane(fg), plane(fh), sw) for (sg, fg, fh, sw) in G["pieces"]]
                G2["Fnorm"] = G["Fnorm"].to(rdt)
                other._gen[XY] = G2
        if getattr(self, "_mv", None) is not None:
            # one stacked allocation again: evenly spaced planes let oa_qe_mv run every estimator's divergence in one launch
            keys = list(self._mv[1])
            stack = other.eng.hcreal(len(keys))
            for i, k in enumerate(keys):
                stack[i].copy_(self._mv[1][k])
            other._mv = (self._mv[0], {k: stack[i] for i, k in enumerate(keys)})
        other.AL, other.Nlkk = dict(self.AL), dict(self.Nlkk)
        other._work = None
        other._rwork = None
        other._racc = None
        other._acc = None
        other._bins = None
        return other

    # ---- data plumbing -----------------------------------------------------------------
    def _as_hc(self, x, alreadyFTed):
        """Map / FT in any accepted container -> hc tensor of the run precision."""
        torch = _torch()
        e = self.eng
        if isinstance(x, HalfPlane):
            t = x.t
            if t.dtype != e.cdt:
                t = t.to(e.cdt)
            return t.contiguous(), "half"
        if alreadyFTed:
            kind = "np" if isinstance(x, np.ndarray) else "torch"
            return e.full_to_hc(e.to_complex(x)), kind
        kind = "np" if isinstance(x, np.ndarray) else "torch"
        return e.rfft(e.to_real(x)), kind

    def _buffers(self):
        if self._work is None:
            e = self.eng
            self._work = dict(G=(e.hc(), e.hc(), e.hc()), C=(e.hc(), e.hc(), e.hc()), P=(e.hc(), e.hc()))
        return self._work

    def _real_buffers(self):
        if getattr(self, "_rwork", None) is None:
            e = self.eng
            self._rwork = (e.real(), e.real(), e.real())
        return self._rwork

    # ---- one-call C-ABI path (include/orphics_amd.h, "one-call entries") -------------------------------------
    def _bind(self):
        """Tell this estimator's plan its TT filters / active region / row grid (plans are shared per geometry:
        rebound whenever another estimator used the plan in between; pointer stores only)."""
        from ._lib import check
        from .engine import _ptr
        e = self.eng
        e._ordered()                      # the plan's work planes are single-buffered (Engine._ordered)
        if getattr(e, "_pipe_owner", None) is not self._token:
            FG, FH, Fn = self._F["TT"]
            wl, wk = self._W["TT"]
            rl, rk = self._R["TT"]
            check(e.lib.oa_plan_set_filters(e.plan, _ptr(FG), _ptr(FH), _ptr(Fn), int(wl), int(wk), int(rl), int(rk), int(self.mrow)))
            check(e.lib.oa_plan_set_col_grid(e.plan, int(self.mcol)))
            e._pipe_owner = self._token
            e._bins_owner = None
        return e

    def bind_bins(self, ids_hc, nids, norm):
        """Radial bins of the one-call Monte-Carlo entries (``tt_moments``, ``mc.GaussianN0MonteCarlo``): int32 ids
        on the hc grid (``Engine.modl_digitize(edges, half=True)``), ``nids = len(edges) + 1``, ``norm = area/Npix^2``."""
        cur = getattr(self, "_bins", None)
        if cur is not None and cur[0] is ids_hc and cur[1] == int(nids) and cur[2] == float(norm):
            return self                   # unchanged: the plan keeps its (whole-plane) mode counts
        self._bins = (ids_hc, int(nids), float(norm))
        self.eng._bins_owner = None
        return self

    def _bind_bins(self):
        from ._lib import check
        from .engine import _ptr, _stream
        e = self._bind()
        if getattr(e, "_bins_owner", None) is not self._token:
            if getattr(self, "_bins", None) is None:
                raise RuntimeError("call bind_bins(ids, nids, norm) first")
            ids, nids, norm = self._bins
            check(e.lib.oa_plan_set_bins(e.plan, _ptr(ids), nids, norm, _stream()))
            e._bins_owner = self._token
        return e

    def tt_moments(self, tmap, n, S, C):
        """One Monte-Carlo step in ONE C-ABI call (``oa_qe_tt_moments``): real device map -> kappa_hat (plan-owned
        plane) -> binned auto-power -> n += 1, S += b, C += b b^T on the device accumulators (int64[1], f64[d],
        f64[d,d], d = nids - 2)."""
        from ._lib import check
        from .engine import _ptr, _stream
        e = self._bind_bins()
        e._chk(tmap, "real")
        check(e.lib.oa_qe_tt_moments(e.plan, _ptr(tmap), _ptr(n), _ptr(S), _ptr(C), _stream()))

    def tt_moments2(self, tmap0, tmap1, n, S, C):
        """Two Monte-Carlo steps in one C-ABI call (``oa_qe_tt_moments2``): the same accumulations as two
        :meth:`tt_moments` calls; on the column-grid path the two maps share every launch behind their row transforms."""
        from ._lib import check
        from .engine import _ptr, _stream
        e = self._bind_bins()
        e._chk(tmap0, "real"); e._chk(tmap1, "real")
        check(e.lib.oa_qe_tt_moments2(e.plan, _ptr(tmap0), _ptr(tmap1), _ptr(n), _ptr(S), _ptr(C), _stream()))

    def bin_counts(self):
        """int64[nids] mode counts per bin over the whole plane (taken by ``oa_plan_set_bins``)."""
        import ctypes
        e = self._bind_bins()
        torch = _torch()
        nids = self._bins[1]
        out = torch.empty(nids, dtype=torch.int64, device=e.device)
        src = e.lib.oa_plan_bin_counts(e.plan)
        from ._lib import check
        from .engine import _ptr, _stream
        check(e.lib.oa_memcpy(_ptr(out), ctypes.c_void_p(src), nids * 8, 3, _stream()))
        return out

    def _qe_tt(self, tmap, kX, kY, out):
        from ._lib import check
        from .engine import _ptr, _stream, mark_dirty, owned_clean_region, set_clean_region
        e = self._bind()
        wk, rk = self._W["TT"][1], self._R["TT"][1]
        if out is None:
            out, zero = e.hc(), 0                                  # zero-initialised
        else:
            e._chk(out, "hc")
            zero = 1 if ((wk or rk) and owned_clean_region(out) != (wk, rk)) else 0
        check(e.lib.oa_qe_tt(e.plan, _ptr(tmap), _ptr(kX), _ptr(kY), _ptr(out), zero, _stream()))
        mark_dirty(out)
        if wk or rk:
            set_clean_region(out, (wk, rk))
        return out

    def reconstruct_tt_hc(self, kX, kY=None, out=None, fused=True):
        """Device-native TT reconstruction: hc tensors in, kappa_hat DFT (hc) out.

        fused=True (default): ONE C-ABI call (``oa_qe_tt``): legs -> 3 inverse column transforms -> ONE fused
        row-stage kernel (3 C2R + 2 products + 2 R2C in LDS) -> 2 forward column transforms -> divergence.
        fused=False: the modular sequence of public C-ABI calls (3 C2R, 2 products, 2 R2C).

        With ``prune`` (default) only the first ``leg_cols`` columns of kX / kY are read and only the first
        ``kappa_cols`` columns of the result are computed; the remaining columns of ``out`` are zero-filled (on
        every call, unless ``out`` came from :meth:`new_output`)."""
        e = self.eng
        if fused and e.pow2:
            e._chk(kX, "hc")
            if kY is not None and kY is not kX:
                e._chk(kY, "hc")
            else:
                kY = None
            return self._qe_tt(None, kX, kY, out)
        kY = kX if kY is None else kY
        FG, FH, Fn = self._F["TT"]
        w = self._buffers()
        Gx, Gy, H = e.qe_legs(kX, kY, FG, FH, out=w["G"])
        gx, gy, h = self._real_buffers()
        e.irfft(Gx, out=gx); e.irfft(Gy, out=gy); e.irfft(H, out=h)
        e.mul_real(gx, h, out=gx)
        e.mul_real(gy, h, out=gy)
        Px, Py = w["P"]
        e.rfft(gx, out=Px); e.rfft(gy, out=Py)
        return e.qe_div(Px, Py, Fn, out=out)

    def reconstruct_tt_from_map(self, tmap, out=None):
        """TT reconstruction straight from a real device map (both legs from it) in ONE C-ABI call: the map's
        transform is consumed inside the fused leg kernel and never written.  Same result as
        ``reconstruct_tt_hc(eng.rfft(tmap))``."""
        e = self.eng
        if not e.pow2:                       # chirp-z sizes: modular chain of public calls
            return self.reconstruct_tt_hc(e.rfft(tmap), out=out)
        e._chk(tmap, "real")
        return self._qe_tt(tmap, None, None, out)

    def tt_pairs(self, ksplits, out=None, owned=False):
        """TT reconstructions of every ordered pair of maps in ONE C-ABI call (``oa_qe_tt_splits``): ``ksplits`` = n hc
        transforms; returns an (n, n, Ny, kp) complex tensor K with K[i, j] = QE(X leg from map i, Y leg from map j).
        The three filtered leg planes of each map are transformed once (n leg stages, not n^2)."""
        import ctypes
        torch = _torch()
        from ._lib import check
        from .engine import _stream, mark_dirty
        e = self._bind()
        n = len(ksplits)
        for k in ksplits:
            e._chk(k, "hc")
        zero = 0
        if out is None:
            out = torch.zeros((n, n, e.ny, e.kp), dtype=e.cdt, device=e.device)
        else:
            if tuple(out.shape) != (n, n, e.ny, e.kp) or out.dtype != e.cdt or not out.is_contiguous():
                raise ValueError("tt_pairs: out must be a contiguous (n, n, Ny, kp) complex tensor of the estimator's precision")
            zero = 0 if owned else 1          # owned: a block this call sequence zero-filled once and nothing else writes
        ins = (ctypes.c_void_p * n)(*[k.data_ptr() for k in ksplits])
        outs = (ctypes.c_void_p * (n * n))(*[out[i, j].data_ptr() for i in range(n) for j in range(n)])
        check(e.lib.oa_qe_tt_splits(e.plan, n, ins, outs, zero, _stream()))
        mark_dirty(out)
        return out

    def kappa_from_map(self, XY, T2DData, E2DData=None, B2DData=None, T2DDataY=None, E2DDataY=None, B2DDataY=None,
                       alreadyFTed=False, returnFt=False):
        """qest.kappa_from_map (lensing.py:973-976; notebook cell 4).  The X
        (gradient) and Y legs may be different maps (SplitLensing)."""
        fields = {"T": (T2DData, T2DDataY), "E": (E2DData, E2DDataY), "B": (B2DData, B2DDataY)}
        X, Y = XY[0], XY[1]
        if fields[X][0] is None or (fields[Y][0] is None and fields[Y][1] is None):
            raise ValueError("estimator %s needs the %s and %s data" % (XY, X, Y))
        ysrc = fields[Y][1] if fields[Y][1] is not None else fields[Y][0]
        xsrc = fields[X][0]
        if XY == "TT" and ysrc is xsrc and not alreadyFTed and not isinstance(xsrc, HalfPlane):
            # both legs from one real map: its transform is consumed inside the fused leg kernel
            kind = "np" if isinstance(xsrc, np.ndarray) else "torch"
            kft = self.reconstruct_tt_from_map(self.eng.to_real(xsrc))
            kX = kY = None
        else:
            kX, kind = self._as_hc(xsrc, alreadyFTed)
            kY = kX if (ysrc is xsrc) else self._as_hc(ysrc, alreadyFTed)[0]
        if kX is None:
            pass
        elif XY == "TT":
            kft = self.reconstruct_tt_hc(kX, kY)
        else:
            kft = self.reconstruct_hc(XY, kX, kY)
        e = self.eng
        if returnFt:
            if kind == "half":
                return HalfPlane(kft, e)
            full = e.hc_to_full(kft)
            return full.cpu().numpy() if kind == "np" else full
        rec = e.irfft(kft)
        return rec.cpu().numpy() if kind == "np" else rec

    # ---- general separable estimators (TE, EE, EB, TB; TT also available for cross-checks) -------
    # term = (coef, p, A, B, trig): coef * (L.l_p) * A(l1) * B(l2) * trig(a2 - a1)   (oracle/qe_oracle.py)
    @staticmethod
    def _terms(XY):
        if XY == "TT":
            return ([(1., 1, "wTT", "iT", "1")], [(1., 1, "cTT", "1", "1"), (1., 2, "1", "cTT", "1")])
        if XY == "EE":
            return ([(1., 1, "wEE", "iE", "cos")], [(1., 1, "cEE", "1", "cos"), (1., 2, "1", "cEE", "cos")])
        if XY == "EB":
            return ([(1., 1, "wEE", "iB", "sin")], [(1., 1, "cEE", "1", "sin"), (1., 2, "1", "cBB", "sin")])
        if XY == "TB":
            return ([(1., 1, "wTE", "iB", "sin")], [(1., 1, "cTE", "1", "sin")])
        if XY == "TE":
            return ([(1., 1, "wTE", "iE", "cos"), (1., 2, "iT", "wET", "1")],
                    [(1., 1, "cTE", "1", "cos"), (1., 2, "1", "cTE", "1")])
        raise ValueError("unknown estimator %r" % (XY,))

    @staticmethod
    def _trig_product(t1, t2):
        key = tuple(sorted((t1, t2)))
        return {("1", "1"): [(1., 0, "1")], ("1", "cos"): [(1., 1, "cos")], ("1", "sin"): [(1., 1, "sin")],
                ("cos", "cos"): [(.5, 0, "1"), (.5, 2, "cos")], ("sin", "sin"): [(.5, 0, "1"), (-.5, 2, "cos")],
                ("cos", "sin"): [(.5, 2, "sin")]}[key]

    def _planes(self):
        """Host half-plane filter / spectrum planes by name (beam-deconvolved field weights)."""
        if getattr(self, "_P", None) is None:
            b2 = self.beam ** 2
            ct = {"T": self.cl_len["TT"] + _safe_div(self.noise["T"], b2),
                  "E": self.cl_len["EE"] + _safe_div(self.noise["P"], b2),
                  "B": self.cl_len["BB"] + _safe_div(self.noise["P"], b2)}
            m = {"T": self.mask["T"], "E": self.mask["P"], "B": self.mask["P"]}
            P = {"1": np.ones_like(self.modl_h)}
            for X in "TEB":
                P["i" + X] = _safe_div(np.ones_like(ct[X]), ct[X]) * m[X]
            gm = {X: m[X].copy() for X in "TEB"}
            if self.grad_cut is not None:
                for X in "TEB":
                    gm[X][self.modl_h > self.grad_cut] = 0
            P["wTT"] = _safe_div(self.cl_grad["TT"], ct["T"]) * gm["T"]
            P["wEE"] = _safe_div(self.cl_grad["EE"], ct["E"]) * gm["E"]
            P["wTE"] = _safe_div(self.cl_grad["TE"], ct["T"]) * gm["T"]
            P["wET"] = _safe_div(self.cl_grad["TE"], ct["E"]) * gm["E"]
            for k in ("TT", "EE", "BB", "TE"):
                P["c" + k] = self.cl_grad[k]
            self._P, self._ct = P, ct
        return self._P

    def _real_of(self, build, cache, key):
        """irfft (f64 kernels) of a Hermitian half-plane DEVICE array ``build()``, cached by description (the array is only built
        when the description is new: most of the ~400 factors of a five-estimator set-up repeat)."""
        if key in cache:
            return cache[key]
        e = self.eng64
        k = e.hc()
        k[:, :self.nxh + 1] = build()
        r = e.irfft(k)
        cache[key] = r
        return r

    def _sum_gf(self, gterms, fterms):
        """(1/Area) sum_l1 g(l1,l2) f(l1,l2) on the half plane via real-space products (f64 kernels).
        Every factor carries exactly two l-components in total, each written as (i l_j) so all planes are
        Hermitian: conv = i^-2 DFT[u v]/a = -DFT[u v]/a; products accumulate per (j,k) class.
        The factor planes (filter products x (i l_j) x cos / sin(m angle)) are formed ON THE DEVICE in complex128: in NumPy on the
        host they were 150 of the 180 s of an 8192^2 five-estimator set-up (tools/profile_setup.py)."""
        torch = _torch()
        e = self.eng64
        dev = e.device
        P = self._planes()
        lyd, lxd = self.ly.copy(), self.lxh.copy()
        lyd[e.ny // 2] = 0.0
        if self.nxh < lxd.size:
            lxd[self.nxh] = 0.0
        # i l_x, i l_y as complex device planes (broadcast views)
        comp = (1j * torch.as_tensor(lxd, dtype=torch.float64, device=dev)[None, :],
                1j * torch.as_tensor(lyd, dtype=torch.float64, device=dev)[:, None])
        ang = torch.as_tensor(self.ang_h, dtype=torch.float64, device=dev) if hasattr(self, "ang_h") else None
        cache, dplane, dprod, dtrig = {}, {}, {}, {}
        S = {"xx": None, "yy": None, "xy": None}

        def tag(x):
            return x if isinstance(x, str) else ("arr", id(x))

        def plane(x):                                   # host plane (by name or array) -> float64 device plane, once per call
            t = tag(x)
            if t not in dplane:
                dplane[t] = torch.as_tensor(np.ascontiguousarray(P[x] if isinstance(x, str) else x), dtype=torch.float64, device=dev)
            return dplane[t]

        def prod(a, b):
            t = (tag(a), tag(b))
            if t not in dprod:
                dprod[t] = plane(a) * plane(b)
            return dprod[t]

        def trig(kind, mh):
            if (kind, mh) not in dtrig:
                dtrig[(kind, mh)] = torch.cos(mh * ang) if kind == "c" else torch.sin(mh * ang)
            return dtrig[(kind, mh)]

        def add(cls, coef, u, v):
            pr = e.mul_real(u, v)
            S[cls] = e.axpby(pr, pr, coef, 0.0) if S[cls] is None else e.axpby(S[cls], pr, 1.0, coef)

        for (cg, p, Ag, Bg, tg) in gterms:
            for (cf, q, Af, Bf, tf) in fterms:
                kA, kB = (tag(Ag), tag(Af)), (tag(Bg), tag(Bf))
                for (ct_, mh, kind) in self._trig_product(tg, tf):
                    for j in range(2):
                        for k in range(2):
                            cls = "xx" if (j, k) == (0, 0) else ("yy" if (j, k) == (1, 1) else "xy")
                            fu, fv = [], []
                            (fu if p == 1 else fv).append(j)
                            (fu if q == 1 else fv).append(k)

                            def build(x, y, comps, tr):
                                def go():
                                    arr = prod(x, y).to(torch.complex128)
                                    for cidx in comps:
                                        arr = arr * comp[cidx]
                                    if tr is not None:
                                        arr = arr * trig(tr, mh)
                                    return arr
                                return go

                            coef = -cg * cf * ct_   # i^-2
                            if kind == "1":
                                u = self._real_of(build(Ag, Af, fu, None), cache, (kA, tuple(fu), None, 0))
                                v = self._real_of(build(Bg, Bf, fv, None), cache, (kB, tuple(fv), None, 0))
                                add(cls, coef, u, v)
                            else:
                                uc = self._real_of(build(Ag, Af, fu, "c"), cache, (kA, tuple(fu), "c", mh))
                                us = self._real_of(build(Ag, Af, fu, "s"), cache, (kA, tuple(fu), "s", mh))
                                vc = self._real_of(build(Bg, Bf, fv, "c"), cache, (kB, tuple(fv), "c", mh))
                                vs = self._real_of(build(Bg, Bf, fv, "s"), cache, (kB, tuple(fv), "s", mh))
                                if kind == "cos":      # cos(m(a2-a1)) = c1 c2 + s1 s2
                                    add(cls, coef, uc, vc); add(cls, coef, us, vs)
                                else:                  # sin(m(a2-a1)) = s2 c1 - c2 s1
                                    add(cls, coef, uc, vs); add(cls, -coef, us, vc)
        del dplane, dprod, dtrig
        zero = torch.zeros((e.ny, e.nx), dtype=e.rdt, device=e.device)
        A_, B_, C_ = [e.rfft(S[c] if S[c] is not None else zero) for c in ("xx", "yy", "xy")]
        C2 = e.hc()
        C2.copy_(C_ * 0.5)
        one_k = e.hc()
        one_k[:, :self.nxh + 1] = 1.0
        F_one = self._hcreal(e, np.ones_like(self.modl_h))
        ux = e.qe_div(A_, C2, F_one)
        uy = e.qe_div(C2, B_, F_one)
        tot = e.qe_div(ux, uy, F_one)           # = -(lx^2 A + lx ly C + ly^2 B)
        Rk = e.f2power(tot, one_k, -1.0 / self.geom.pixarea)
        return Rk.cpu().numpy()[:, :self.nxh + 1].astype(np.float64)

    def _setup_general(self, XY):
        if XY in self._gen:
            return self._gen[XY]
        if getattr(self, "_fdev", None) is None:
            self._fdev = {}
        if not self.pol and XY != "TT":
            raise ValueError("construct the estimator with pol=True for %s" % XY)
        P = self._planes()
        g, f = self._terms(XY)
        R = self._sum_gf(g, f)
        AL = _safe_div(np.ones_like(R), R)
        L = self.modl_h
        X, Y = XY[0], XY[1]
        cross = {"TT": self._ct["T"], "EE": self._ct["E"], "BB": self._ct["B"], "TE": self.cl_len["TE"]}
        cXY = cross.get(X + Y, cross.get(Y + X, np.zeros_like(L)))
        gp = [(c, p, P[A] * self._ct[X] if p == 1 else P[A] * self._ct[X], P[B] * self._ct[Y], t) for (c, p, A, B, t) in g]
        gs = [(c * (-1. if t == "sin" else 1.), 3 - p, P[B] * cXY, P[A] * cXY, t) for (c, p, A, B, t) in g]
        n1 = self._sum_gf(g, gp)
        n2 = self._sum_gf(g, gs) if np.any(cXY) else 0.0
        self.AL[XY] = AL
        self.Nlkk[XY] = (L * (L + 1.)) ** 2 / 4. * AL ** 2 * (n1 + n2)
        Fnorm = -(L * (L + 1.) / 2.) * AL * self.mask_K
        # device filter planes per weight term and trig piece: (sign, FG, FH, swap_legs)
        c, s_ = np.cos(self.ang_h), np.sin(self.ang_h)
        pieces, hostf = [], []
        for (cg, p, A, B, trig) in g:
            FGh, FHh = (P[A], P[B]) if p == 1 else (P[B], P[A])
            FGh, FHh = _safe_div(FGh, self.beam), _safe_div(FHh, self.beam)
            if trig == "1":
                tl = [(1., None, None)]
            elif trig == "cos":
                tl = [(1., c, c), (1., s_, s_)]
            else:  # sin(a2 - a1): gradient leg at l1 (p==1) or at l2 (p==2)
                tl = [(1., c, s_), (-1., s_, c)] if p == 1 else [(1., s_, c), (-1., c, s_)]
            nG, nH = (A, B) if p == 1 else (B, A)
            for (sg, tg, th) in tl:
                fg = FGh if tg is None else FGh * tg
                fh = FHh if th is None else FHh * th
                # one device plane per distinct (weight name, trig factor): estimators that use the same filtered field
                # hand the SAME plane object to the C-ABI, which then transforms that field once (oa_qe_mv)
                tagG = (nG, None if tg is None else ("c" if tg is c else "s"))
                tagH = (nH, None if th is None else ("c" if th is c else "s"))
                for tag, arr in ((tagG, fg), (tagH, fh)):
                    if tag not in self._fdev:
                        self._fdev[tag] = self._hcreal(self.eng, arr)
                pieces.append((cg * sg, self._fdev[tagG], self._fdev[tagH], p == 2))
                hostf += [fg, fh]
        self._gen[XY] = dict(pieces=pieces, Fnorm=self._hcreal(self.eng, Fnorm), R=R,
                             wl=self._support_cols(*hostf), wk=self._support_cols(Fnorm),
                             rl=self._support_rows(*hostf), rk=self._support_rows(Fnorm))
        return self._gen[XY]

    def reconstruct_hc(self, XY, kX, kY, out=None, norm=None, accumulate=False):
        """General estimator on hc tensors: kX = DFT of field XY[0], kY = DFT of field XY[1].
        ``norm`` overrides the divergence/normalisation plane (MV weights), ``accumulate`` adds into ``out``."""
        e = self.eng
        if not e.pow2:
            return self._reconstruct_hc_modular(XY, kX, kY, out, norm, accumulate)
        import ctypes
        from ._lib import check
        from .engine import _ptr, _stream, mark_dirty, owned_clean_region, set_clean_region
        G = self._setup_general(XY)
        e._chk(kX, "hc"); e._chk(kY, "hc")
        # active columns: legs from this estimator's filters; kappa from its normalisation (an external
        # ``norm`` plane -- MV weights -- is bounded by the kappa mask)
        wl, rl = G["wl"], G["rl"]
        if norm is not None and getattr(self, "_wK", None) is None:
            self._wK = (self._support_cols(self.mask_K), self._support_rows(self.mask_K))
        wk, rk = (G["wk"], G["rk"]) if norm is None else self._wK
        if "c_args" not in G:                 # per-piece argument arrays of oa_qe_pol (host side, built once)
            n = len(G["pieces"])
            G["c_args"] = (n, (ctypes.c_double * n)(*[float(pc[0]) for pc in G["pieces"]]),
                           (ctypes.c_void_p * n)(*[pc[1].data_ptr() for pc in G["pieces"]]),
                           (ctypes.c_void_p * n)(*[pc[2].data_ptr() for pc in G["pieces"]]),
                           (ctypes.c_int * n)(*[1 if pc[3] else 0 for pc in G["pieces"]]))
        n, signs, fgs, fhs, swaps = G["c_args"]
        if out is None:
            out, zero = e.hc(), 0
        else:
            e._chk(out, "hc")
            zero = 1 if ((wk or rk) and not accumulate and owned_clean_region(out) != (wk, rk)) else 0
        Fn = G["Fnorm"] if norm is None else norm
        e._ordered()
        check(e.lib.oa_plan_set_col_grid(e.plan, int(self.mcol)))       # plans are shared per geometry: policy per call
        # one estimator through oa_qe_mv (nest = 1): the same pieces as oa_qe_pol, with all distinct leg planes of the estimator
        # in ONE inverse pass-1 launch and one pass-2 launch (a piece pair shares its cos / sin filtered fields)
        one = ctypes.c_void_p * 1
        check(e.lib.oa_qe_mv(e.plan, 1, (ctypes.c_int * 1)(n), signs, fgs, fhs, swaps, one(kX.data_ptr()), one(kY.data_ptr()),
                             one(Fn.data_ptr()), _ptr(out), 1 if accumulate else 0, int(wl), int(wk), int(rl), int(rk), int(self.mrow), zero,
                             _stream()))
        mark_dirty(out)
        if wk or rk:
            set_clean_region(out, (wk, rk))
        return out

    def _reconstruct_hc_modular(self, XY, kX, kY, out=None, norm=None, accumulate=False):
        """The same estimator through the modular public calls only (map sides that are not powers of two, where
        the fused kernels do not exist): per separable piece legs -> 3 C2R -> 2 real-space products, summed over
        the pieces in real space (the forward transform is linear), then 2 R2C and the divergence."""
        e = self.eng
        G = self._setup_general(XY)
        w = self._buffers()
        gx, gy, h = self._real_buffers()
        if getattr(self, "_racc", None) is None:
            self._racc = (e.real(), e.real())
        ax, ay = self._racc
        for i, (sign, FG, FH, swap) in enumerate(G["pieces"]):
            kg, kh = (kY, kX) if swap else (kX, kY)
            Gx, Gy, H = e.qe_legs(kg, kh, FG, FH, out=w["G"])
            e.irfft(Gx, out=gx); e.irfft(Gy, out=gy); e.irfft(H, out=h)
            e.mul_real(gx, h, out=gx)
            e.mul_real(gy, h, out=gy)
            if i == 0:
                e.axpby(gx, gx, float(sign), 0.0, out=ax)
                e.axpby(gy, gy, float(sign), 0.0, out=ay)
            else:
                e.axpby(gx, ax, float(sign), 1.0, out=ax)
                e.axpby(gy, ay, float(sign), 1.0, out=ay)
        Px, Py = w["P"]
        e.rfft(ax, out=Px); e.rfft(ay, out=Py)
        return e.qe_div(Px, Py, G["Fnorm"] if norm is None else norm, out=out, accumulate=accumulate)

    # ---- minimum-variance combination (BASELINE config 3) -------------------------------------------
    def mv_weights(self, estimators=("TT", "TE", "EE", "EB", "TB")):
        """Per-mode inverse-noise weights w_a = N_a^-1 / sum_b N_b^-1 (diagonal approximation: the
        cross-estimator covariances are neglected, as in the reference notebooks' MV curves) and the
        resulting N_L^kk,MV = 1 / sum_b N_b^-1."""
        for XY in estimators:
            self._setup_general(XY) if XY != "TT" or XY not in self.Nlkk else None
        inv = {XY: _safe_div(np.ones_like(self.modl_h), self.Nlkk[XY]) for XY in estimators}
        tot = sum(inv.values())
        w = {XY: _safe_div(inv[XY], tot) for XY in estimators}
        self.Nlkk["MV"] = _safe_div(np.ones_like(tot), tot)
        return w

    def reconstruct_mv_hc(self, kT, kE, kB, estimators=("TT", "TE", "EE", "EB", "TB"), out=None, fused=True):
        """kappa_hat^MV = sum_a w_a kappa_hat^a, accumulated in the divergence kernel.  fused (default): one ``oa_qe_mv``
        call in which every distinct filtered field is transformed once (TT+TE+EE+EB+TB: 17 leg planes, not 30);
        fused=False: one ``oa_qe_pol`` call per estimator."""
        e = self.eng
        key = tuple(estimators)
        if getattr(self, "_mv", None) is None or self._mv[0] != key:
            w = self.mv_weights(estimators)
            L = self.modl_h
            planes = {}
            stack = e.hcreal(len(estimators))          # ONE allocation: evenly spaced planes let oa_qe_mv run every estimator's
            for i, XY in enumerate(estimators):        # divergence in one launch
                AL = self.AL[XY]
                stack[i].copy_(self._hcreal(e, -(L * (L + 1.) / 2.) * AL * self.mask_K * w[XY]))
                planes[XY] = stack[i]
            self._mv = (key, planes)
        f = {"T": kT, "E": kE, "B": kB}
        if not e.pow2 or not fused:
            out = e.hc() if out is None else out
            for i, XY in enumerate(estimators):
                self.reconstruct_hc(XY, f[XY[0]], f[XY[1]], out=out, norm=self._mv[1][XY], accumulate=(i > 0))
            return out
        # ONE C-ABI call (oa_qe_mv): distinct filtered fields transformed once, estimators accumulated in the divergence
        import ctypes
        from ._lib import check
        from .engine import _ptr, _stream, mark_dirty, owned_clean_region, set_clean_region
        if len(self._mv) < 3:
            G = [self._setup_general(XY) for XY in estimators]
            pcs = [pc for g_ in G for pc in g_["pieces"]]
            n = len(pcs)
            if getattr(self, "_wK", None) is None:
                self._wK = (self._support_cols(self.mask_K), self._support_rows(self.mask_K))
            # common active region: filters vanish outside their own, so the widest one is exact for all (0 = everything)
            widest = lambda vals: 0 if any(v == 0 for v in vals) else max(vals)      # noqa: E731
            args = dict(ne=len(G), npieces=(ctypes.c_int * len(G))(*[len(g_["pieces"]) for g_ in G]),
                        signs=(ctypes.c_double * n)(*[float(pc[0]) for pc in pcs]),
                        fgs=(ctypes.c_void_p * n)(*[pc[1].data_ptr() for pc in pcs]),
                        fhs=(ctypes.c_void_p * n)(*[pc[2].data_ptr() for pc in pcs]),
                        swaps=(ctypes.c_int * n)(*[1 if pc[3] else 0 for pc in pcs]),
                        fns=(ctypes.c_void_p * len(G))(*[self._mv[1][XY].data_ptr() for XY in estimators]),
                        wl=widest([g_["wl"] for g_ in G]), rl=widest([g_["rl"] for g_ in G]))
            self._mv = (self._mv[0], self._mv[1], args)
        a = self._mv[2]
        for XY in estimators:
            e._chk(f[XY[0]], "hc"); e._chk(f[XY[1]], "hc")
        wk, rk = self._wK
        if out is None:
            out, zero = e.hc(), 0
        else:
            e._chk(out, "hc")
            zero = 1 if ((wk or rk) and owned_clean_region(out) != (wk, rk)) else 0
        kxs = (ctypes.c_void_p * a["ne"])(*[f[XY[0]].data_ptr() for XY in estimators])
        kys = (ctypes.c_void_p * a["ne"])(*[f[XY[1]].data_ptr() for XY in estimators])
        e._ordered()
        check(e.lib.oa_plan_set_col_grid(e.plan, int(self.mcol)))
        check(e.lib.oa_qe_mv(e.plan, a["ne"], a["npieces"], a["signs"], a["fgs"], a["fhs"], a["swaps"], kxs, kys, a["fns"], _ptr(out), 0,
                             int(a["wl"]), int(wk), int(a["rl"]), int(rk), int(self.mrow), zero, _stream()))
        mark_dirty(out)
        if wk or rk:
            set_clean_region(out, (wk, rk))
        return out

    # full-plane views of the normalisation (host, float64)
    def _full(self, half):
        Ny, Nx = self.shape[-2:]
        out = np.empty((Ny, Nx))
        out[:, :self.nxh + 1] = half
        idx = (-np.arange(Ny)) % Ny
        out[:, self.nxh + 1:] = half[idx][:, 1:self.nxh][:, ::-1]
        return out

    def N_kappa(self, XY="TT"):
        return self._full(self.Nlkk[XY])


def qest(shape, wcs, theory, **kwargs):
    """``lensing.qest(...)`` constructor name used by the reference notebooks (tutorials/tt_verification.ipynb cell 3)."""
    return Estimator(shape, wcs, theory, **kwargs)


class SplitLensing(object):
    """lensing.py:959-1003: split-based 4-point estimator built from QE calls
    with distinct X / Y legs."""

    def __init__(self, shape, wcs, qest, XY="TT"):
        self.fc = maps.FourierCalc(shape, wcs)
        self.qest = qest
        self.est = XY

    def qpower(self, k1, k2):
        return self.fc.f2power(k1, k2)

    def qfrag(self, a, b):
        if self.est == 'TT':
            return self.qest.kappa_from_map(self.est, T2DData=a, T2DDataY=b, alreadyFTed=True, returnFt=True)
        raise NotImplementedError("SplitLensing: the reference's EE branch is marked wrong (lensing.py:975)")

    def cross_estimator(self, ksplits):
        """lensing.py:980-1003: the split-based 4-point estimate of the kappa power, 2-D.

        The QE is bilinear in its two legs, so everything the reference evaluates -- QE(s, s), QE(m_i, s), QE(s, m_i)
        with s the mean split -- is a mean of the n^2 pairwise reconstructions K_ij = QE(m_i, m_j):

            kc  = mean_ij K_ij - sum_i K_ii / n^2,   kic = sum_j (K_ij + K_ji) / (2n) - K_ii / n,   kij = (K_ij + K_ji) / 2
            result = (n^4 P(kc) - 4 n^2 sum_i P(kic) + 4 sum_{i<j} P(kij)) / (n (n-1)(n-2)(n-3))

        n^2 reconstructions instead of the reference's 1 + 3n + n(n-1).  With this package's TT :class:`Estimator`
        (power-of-two sides, 4 <= n <= 8) the whole matrix is ONE ``oa_qe_tt_splits`` call that transforms each split's
        leg planes once, and the combination is one ``oa_split_cross_power`` launch (f64 arithmetic per mode).  Any
        other ``qest`` object (duck-typed ``kappa_from_map``) goes through ``qfrag`` pair by pair."""
        half = isinstance(ksplits, HalfPlane)
        if half:
            splits = [ksplits[i] for i in range(ksplits.t.shape[0])]
        else:
            arr = np.asanyarray(ksplits)
            splits = [arr[i] for i in range(arr.shape[0])]
        n = len(splits)
        q = self.qest
        if self.est == "TT" and isinstance(q, Estimator) and q.eng.pow2 and 4 <= n <= 8:
            return self._cross_estimator_device(splits, ksplits)
        # generic: pairwise reconstructions through the public qfrag, combined on whatever array type it returns
        val = lambda x: x.t if isinstance(x, HalfPlane) else x      # noqa: E731
        raw = [[self.qfrag(splits[i], splits[j]) for j in range(n)] for i in range(n)]
        K = [[val(x) for x in row] for row in raw]
        like = raw[0][0] if half else None

        def power(x):
            return val(self.qpower(HalfPlane(x, like.eng), HalfPlane(x, like.eng)) if half else self.qpower(x, x))

        fn = float(n)
        diag = sum(K[i][i] for i in range(n))
        kc = sum(K[i][j] for i in range(n) for j in range(n)) / fn ** 2 - diag / fn ** 2
        pic = 0.
        pij = 0.
        for i in range(n):
            kic = sum(K[i][j] + K[j][i] for j in range(n)) / (2. * fn) - K[i][i] / fn
            pic = pic + power(kic)
            for j in range(i + 1, n):
                pij = pij + power((K[i][j] + K[j][i]) / 2.)
        res = (fn ** 4. * power(kc) - 4. * fn ** 2. * pic + 4. * pij) / fn / (fn - 1.) / (fn - 2.) / (fn - 3.)
        return HalfPlane(res, like.eng) if half else res

    def _cross_estimator_device(self, splits, like):
        import ctypes
        from ._lib import check
        from .engine import _ptr, _stream
        q = self.qest
        e = q.eng
        n = len(splits)
        hcs = []
        kind = "half"
        for m in splits:
            k, kind = q._as_hc(m, True)
            hcs.append(k)
        # the n^2 kappa planes live in a block this object owns: zero outside kappa's active region once (4.3 GB of
        # zero-fill per estimate at 8192^2 otherwise), only the active region is rewritten per call
        key = (n, id(q), e.cdt)
        if getattr(self, "_pairs", None) is None or self._pairs[0] != key:
            self._pairs = (key, _torch().zeros((n, n, e.ny, e.kp), dtype=e.cdt, device=e.device))
        K = q.tt_pairs(hcs, out=self._pairs[1], owned=True)
        wk, rk = q._W["TT"][1], q._R["TT"][1]
        out = e.hcreal()                              # zero outside kappa's active region, like every K_ij
        planes = (ctypes.c_void_p * (n * n))(*[K[i, j].data_ptr() for i in range(n) for j in range(n)])
        check(e.lib.oa_split_cross_power(e.code, n, planes, _ptr(out), float(self.fc.normfact), e.ny, e.kp, int(wk), int(rk), _stream()))
        if kind == "half":
            return HalfPlane(out, e)
        full = e.hcreal_to_full(out)
        return full.cpu().numpy() if kind == "np" else full


# ---- kappa -> phi -> deflection, flat-sky Taylens -------------------------------
def _eng_for(x, shape):
    from .engine import precision_of
    return maps._engine(shape, precision_of(x, default="f32"))


class FlatLenser(object):
    """kappa_to_phi / alpha_from_kappa / flat_taylens (lensing.py:651-665, 443-454, 395-440) on the device.
    Coordinates are signed (x decreases with pixel index for standard CAR), so the pixel shift of a
    displacement component is alpha/step with the signed step."""

    def __init__(self, shape, wcs, dtype="f32"):
        self.shape = tuple(shape[-2:])
        self.geom = as_geometry(self.shape, wcs)
        self.eng = maps._engine(self.shape, dtype)
        e = self.eng
        ly, lx = self.geom.laxes()
        e.set_laxes(ly, lx)
        nxh = e.nxh
        ml = np.sqrt(ly[:, None] ** 2 + lx[None, :nxh + 1] ** 2)
        with np.errstate(divide="ignore", invalid="ignore"):
            f = np.nan_to_num(2. / ml / (ml + 1.))
        f[ml < 2.] = 0.                                   # fkappa_to_fphi, lensing.py:662-665
        torch = _torch()
        self._fphi = e.hcreal(); self._fphi[:, :nxh + 1] = torch.as_tensor(f, dtype=e.rdt, device=e.device)
        self._one = e.hcreal(); self._one[:, :nxh + 1] = 1.0

    def alpha_from_kappa(self, kappa):
        """grad(phi) with phi = IDFT[2 kappa_l/(l(l+1))]: returns (alpha_y, alpha_x) device maps."""
        e = self.eng
        return self.alpha_from_kappa_hc(e.rfft(e.to_real(kappa)))

    def alpha_from_kappa_hc(self, kk):
        """the same from kappa's (unnormalised) hc transform -- a simulation that drew kappa in harmonic space has it already"""
        e = self.eng
        if getattr(self, "_alpha_work", None) is None:       # the three leg planes of the gradient kernel, allocated once (not zero-filled per call)
            self._alpha_work = (e.hc(), e.hc(), e.hc())
        gx, gy, _ = e.qe_legs(kk, kk, self._fphi, self._fphi, out=self._alpha_work)
        return e.irfft(gy), e.irfft(gx)

    def kappa_to_phi(self, kappa):
        e = self.eng
        return e.irfft(e.cmul_real(e.rfft(e.to_real(kappa)), self._fphi))

    def split(self, alpha):
        """Nearest-pixel shifts and sub-pixel remainders of a deflection field (alpha_y, alpha_x).  Pure function of the
        tensors' CONTENTS at call time: nothing is cached (a cache keyed on buffer addresses returned a previous
        realisation's split once the allocator reused the blocks).  Callers that lens several maps by one deflection
        (T, Q, U of a realisation: ``FlatLensingSims.lens_maps``) compute it once and pass ``lens(..., split=...)``."""
        e = self.eng
        ay, ax = alpha
        sx, dx = e.lens_split(ax, self.geom.step_x)
        sy, dy = e.lens_split(ay, self.geom.step_y)
        return sx, sy, dx, dy

    def lens_many(self, imaps, alpha, taylor_order=5, split=None, out=None):
        """flat_taylens (lensing.py:395-440) of several maps -- ``imaps``: (n, Ny, Nx) device tensor, e.g. T, Q, U of one
        realisation -- by ONE deflection, in one C-ABI call (``oa_lens_maps``): the R2C of every map, then the inverse transforms
        of all n * nd derivative fields in three launches (the factor (i lx)^a (i ly)^b rides on the load of the inverse column
        pass: no derivative spectrum in HBM, no Python loop over terms), one gather pass per map.  The work planes belong to
        the plan (n * (1 + nd) planes + one cache-sized chunk: 6.2 GB for n = 3 at 4096^2 float64, order 5; ``release()`` frees
        them)."""
        from ._lib import check
        from .engine import _ptr, _stream
        e = self.eng
        torch = _torch()
        sx, sy, dx, dy = split if split is not None else self.split(alpha)
        src = imaps if (torch.is_tensor(imaps) and imaps.is_cuda and imaps.dtype == e.rdt) else torch.as_tensor(np.asarray(imaps), dtype=e.rdt, device=e.device)
        src = src.contiguous()
        if src.ndim != 3 or tuple(src.shape[-2:]) != (e.ny, e.nx):
            raise ValueError("lens_many: expected (n, %d, %d) maps" % (e.ny, e.nx))
        if out is None:
            out = torch.empty_like(src)
        e._ordered()                                   # the work planes belong to the plan: calls from another stream queue behind
        check(e.lib.oa_lens_maps(e.plan, int(src.shape[0]), _ptr(src), e.ny * e.nx, int(taylor_order), _ptr(sx), _ptr(sy), _ptr(dx), _ptr(dy),
                                 _ptr(out), e.ny * e.nx, _stream()))
        return out

    def lens_many_hc(self, khc, alpha, taylor_order=5, split=None, scale=None, out=None):
        """:meth:`lens_many` from the maps' hc transforms (``khc``: (n, Ny, kp) complex device tensor; map = scale x C2R(khc),
        ``scale`` default 1 / sqrt(Ny Nx): MapGen's unitary draws) -- ``oa_lens_maps_hc``: no forward transform, and the
        undisplaced maps come out of the same batched row launches as their derivatives."""
        from ._lib import check
        from .engine import _ptr, _stream
        e = self.eng
        torch = _torch()
        sx, sy, dx, dy = split if split is not None else self.split(alpha)
        if not (torch.is_tensor(khc) and khc.is_cuda and khc.dtype == e.cdt and khc.ndim == 3 and tuple(khc.shape[-2:]) == (e.ny, e.kp)):
            raise ValueError("lens_many_hc: expected a (n, %d, %d) %s device tensor" % (e.ny, e.kp, e.cdt))
        khc = khc.contiguous()
        if out is None:
            out = torch.empty((khc.shape[0], e.ny, e.nx), dtype=e.rdt, device=e.device)
        sc = 1.0 / float(np.sqrt(e.npix)) if scale is None else float(scale)
        e._ordered()
        check(e.lib.oa_lens_maps_hc(e.plan, int(khc.shape[0]), _ptr(khc), e.ny * e.kp, sc, int(taylor_order), _ptr(sx), _ptr(sy), _ptr(dx),
                                    _ptr(dy), _ptr(out), e.ny * e.nx, _stream()))
        return out

    def release(self):
        """free the plan-owned work planes of ``lens`` / ``lens_many`` (they are reallocated by the next call)"""
        self.eng.release_pools()

    def lens(self, imap, alpha, taylor_order=5, fused=True, split=None):
        """flat_taylens (lensing.py:395-440): T(x + alpha) by nearest-pixel remap + Taylor series in FFT derivatives.
        fused (default): one ``oa_lens_maps`` call (see :meth:`lens_many`); fused=False: one derivative kernel, C2R and gather
        per term through the public fine-grained calls (the first implementation, kept as the cross-check)."""
        from math import factorial
        e = self.eng
        sx, sy, dx, dy = split if split is not None else self.split(alpha)
        src = e.to_real(imap)
        if fused and 1 <= taylor_order <= 8 and (e.pow2 or e.mixed):
            return self.lens_many(src[None], alpha, taylor_order=taylor_order, split=(sx, sy, dx, dy))[0]
        out = e.real()
        e.lens_gather(src, sx, sy, dx, dy, 0, 0, 1.0, out, False)
        k0 = e.rfft(src)
        # D[a][b] = (i lx)^a (i ly)^b k0, built incrementally with the derivative kernel
        row = {(0, 0): k0}
        for n in range(1, taylor_order):
            for b in range(n + 1):
                a = n - b
                if (a, b) in row:
                    continue
                if a > 0:
                    dxk, _, _ = e.qe_legs(row[(a - 1, b)], k0, self._one, self._one)
                    row[(a, b)] = dxk
                else:
                    _, dyk, _ = e.qe_legs(row[(a, b - 1)], k0, self._one, self._one)
                    row[(a, b)] = dyk
            for b in range(n + 1):
                a = n - b
                d = e.irfft(row[(a, b)])
                e.lens_gather(d, sx, sy, dx, dy, a, b, 1.0 / (factorial(a) * factorial(b)), out, True)
            for key in [kk for kk in row if kk[0] + kk[1] < n]:   # only the previous order is needed to go on
                if key != (0, 0):
                    del row[key]
        return out


class FlatLensingSims(object):
    """lensing.py:458-521: CMB / kappa / noise GRF generators, Gaussian beam,
    white noise.  The lensing operation itself (pixell displace_map) is the
    SURVEY.md section 8f-1 'next' row; ``get_sim(skip_lensing=True)`` is the
    Gaussian (N0 / mean-field) simulation used by the Monte-Carlo driver."""

    def __init__(self, shape, wcs, theory, beam_arcmin, noise_uk_arcmin, noise_e_uk_arcmin=None, noise_b_uk_arcmin=None,
                 pol=False, fixed_lens_kappa=None, dtype="f32"):
        if len(shape) < 3 and pol:
            shape = (3,) + tuple(shape)
        self.shape = tuple(shape)
        self.wcs = wcs
        self.geom = as_geometry(shape, wcs)
        if noise_e_uk_arcmin is None:
            noise_e_uk_arcmin = np.sqrt(2.) * noise_uk_arcmin
        if noise_b_uk_arcmin is None:
            noise_b_uk_arcmin = noise_e_uk_arcmin
        self.modlmap = self.geom.modlmap()
        Ny, Nx = self.shape[-2:]
        ps_cmb = power_from_theory(self.modlmap, theory, lensed=False, pol=pol)
        self.mgen = maps.MapGen(self.shape, wcs, ps_cmb, dtype=dtype)
        ps_kk = theory.gCl('kk', self.modlmap).reshape((1, 1, Ny, Nx))
        self.kgen = maps.MapGen(self.shape[-2:], wcs, ps_kk, dtype=dtype)
        self.ps_kk = ps_kk
        self.kbeam = maps.gauss_beam(self.modlmap, beam_arcmin)
        ncomp = 3 if pol else 1
        ps_noise = np.zeros((ncomp, ncomp, Ny, Nx))
        ps_noise[0, 0] = (noise_uk_arcmin * np.pi / 180. / 60.) ** 2.
        if pol:
            ps_noise[1, 1] = (noise_e_uk_arcmin * np.pi / 180. / 60.) ** 2.
            ps_noise[2, 2] = (noise_b_uk_arcmin * np.pi / 180. / 60.) ** 2.
        self.ngen = maps.MapGen(self.shape, wcs, ps_noise, dtype=dtype)
        self.ps_noise = ps_noise
        self._fixed = fixed_lens_kappa is not None
        self.kappa = fixed_lens_kappa
        self.lenser = FlatLenser(self.shape[-2:], wcs, dtype=dtype)
        self.alpha = self.lenser.alpha_from_kappa(self.kappa) if self._fixed else None

    def get_unlensed(self, seed=None):
        return self.mgen.get_map(seed=seed)

    def lens_maps(self, unlensed, alpha, lens_order=5):
        """every component of ``unlensed`` displaced by the deflection ``alpha`` = (alpha_y, alpha_x).  The reference calls
        pixell.lensing.displace_map(order=lens_order) (lensing.py:512); here the FFT-only Taylens of lensing.py:395-440 is
        used (same Taylor order)."""
        torch = _torch()
        sp = self.lenser.split(alpha)                       # once per deflection, shared by every component
        if unlensed.ndim == 2:
            return self.lenser.lens(unlensed, alpha, taylor_order=lens_order, split=sp)
        if (self.lenser.eng.pow2 or self.lenser.eng.mixed) and 1 <= lens_order <= 8:   # all components in ONE oa_lens_maps call
            return self.lenser.lens_many(unlensed, alpha, taylor_order=lens_order, split=sp)
        return torch.stack([self.lenser.lens(unlensed[i].contiguous(), alpha, taylor_order=lens_order, split=sp) for i in range(unlensed.shape[0])])

    def beam_maps(self, lensed):
        """filter_map(lensed, kbeam) (lensing.py:513) with the beam plane resident on the device"""
        cached = getattr(self, "_kbeam_dev", None)         # the beam plane goes to the device once per kbeam OBJECT, not per realisation
        if cached is None or cached[0] is not self.kbeam:
            self._kbeam_dev = (self.kbeam, maps.prepare_filter(self.shape[-2:], self.kbeam, dtype=self.lenser.eng.prec))
        return maps.filter_map(lensed, self._kbeam_dev[1])

    def get_kappa(self, seed=None):
        return self.kgen.get_map(seed=seed, scalar=True)

    def iau_mismatch(self, qest):
        """get_sim_teb rotates Q, U -> E, B with the convention the maps were drawn in (MapGen.get_map: iau=False); an estimator
        set up with the other sign convention needs the observed MAPS (get_sim) and its own rotation"""
        return bool(getattr(qest, "iau", False))

    def get_sim_teb(self, seed_cmb=None, seed_kappa=None, seed_noise=None, lens_order=5):
        """What the verification loop feeds to the estimators (tutorials/tt_verification.ipynb cell 4: ``FourierCalc.iqu2teb`` of
        the observed maps and the transform of the input kappa) WITHOUT taking any transform twice.  ``get_sim`` draws every field
        in harmonic space, inverse-transforms it, and the loop transforms the results forward again; here
          * kappa's drawn transform goes straight to ``alpha_from_kappa_hc`` and is returned as the loop's ``kappa_in`` transform,
          * the beam multiplies the transform of the lensed maps and the noise -- drawn as T, E, B transforms -- is added there
            (per-mode linear operations commute with the transform; the Q, U -> E, B rotation is applied to the beamed signal),
        so a polarised simulation costs 3 R2Cs + (3 + 2 + 3 nd) C2Rs instead of 11 + (12 + 3 nd).  Same realisations as
        ``get_sim`` with the same seeds; T, E, B agree with ``iqu2teb(get_sim(...))`` to rounding below the Nyquist modes.
        Returns (teb: (ncomp, Ny, kp) unnormalised hc transforms, kappa_hc: unnormalised hc transform of the input kappa)."""
        torch = _torch()
        e = self.lenser.eng
        rt = float(np.sqrt(e.npix))                       # MapGen draws are unitary: rfft(map) = sqrt(Npix) x the drawn transform
        assert not self._fixed, "get_sim_teb draws its own kappa"
        cached = getattr(self, "_kbeam_hc", None)
        if cached is None or cached[0] is not self.kbeam:
            self._kbeam_hc = (self.kbeam, e.fullreal_to_hc(e.to_real(self.kbeam)))
        beam = self._kbeam_hc[1]
        pol = len(self.shape) > 2 and self.shape[0] == 3
        if (e.pow2 or e.mixed) and 1 <= lens_order <= 8 and (pol or len(self.shape) == 2):
            # device pipeline: one draw-and-mix pass per field set (oa_grf_mix: covsqrt x white, E, B -> Q, U), the lens operation from
            # the drawn transforms (oa_lens_maps_hc: no transform of the unlensed maps in either direction), and one pass for
            # beam x lensed -> E, B + noise
            kunl = self.mgen.draw_hc(seed_cmb, rot="inverse" if pol else None)
            kin = self.kgen.draw_hc(seed_kappa, scale=rt)[0]
            self.alpha = self.lenser.alpha_from_kappa_hc(kin)
            lensed = self.lenser.lens_many_hc(kunl, self.alpha, taylor_order=lens_order, split=self.lenser.split(self.alpha))
            ks = torch.empty_like(kunl)
            for i in range(lensed.shape[0]):
                e.rfft(lensed[i], out=ks[i])
            teb = self.ngen.draw_hc(seed_noise, rot="forward" if pol else None, inputs=ks, filt=beam, scale=rt, out=ks)
            return teb, kin
        unl = self.get_unlensed(seed_cmb)
        kin = self.kgen.get_map(seed=seed_kappa, scalar=True, harm=True).t * rt
        self.alpha = self.lenser.alpha_from_kappa_hc(kin)
        lensed = self.lens_maps(unl, self.alpha, lens_order)
        pol = lensed.ndim == 3
        planes = [lensed[i] for i in range(lensed.shape[0])] if pol else [lensed]
        ks = [e.cmul_real(e.rfft(p.contiguous()), beam) for p in planes]
        if pol:
            if getattr(self, "_fc_half", None) is None:
                self._fc_half = maps.FourierCalc(self.shape, self.wcs, layout="half")
            c, s = self._fc_half._rot_planes(e, True)
            ks[1], ks[2] = e.rot2(c, s, ks[1], ks[2])
        nk = self.ngen.get_map(seed=seed_noise, harm=True).t
        nk = [nk[i] for i in range(nk.shape[0])] if pol else [nk]
        teb = torch.stack([k.add_(n, alpha=rt) for k, n in zip(ks, nk)])
        return teb, kin

    def get_sim(self, seed_cmb=None, seed_kappa=None, seed_noise=None, lens_order=5, return_intermediate=False,
                skip_lensing=False, cfrac=None):
        """lensing.py:499-521."""
        torch = _torch()
        unlensed = self.get_unlensed(seed_cmb)
        if skip_lensing:
            lensed = unlensed
            kappa = torch.zeros_like(lensed if lensed.ndim == 2 else lensed[0])
        else:
            if not self._fixed:
                kappa = self.get_kappa(seed_kappa)
                self.kappa = kappa
                self.alpha = self.lenser.alpha_from_kappa(kappa)
            else:
                kappa = None
                assert seed_kappa is None
            lensed = self.lens_maps(unlensed, self.alpha, lens_order)
        beamed = self.beam_maps(lensed)
        noise_map = self.ngen.get_map(seed=seed_noise)
        observed = beamed + noise_map
        if return_intermediate:
            return [unlensed, kappa, lensed, beamed, noise_map, observed]
        return observed


# ---- analytic N_L / MV / iterative delensing ------------------------------------
class NlGenerator(object):
    """Call contract of the reference's ``lensing.NlGenerator`` as it survives in
    tutorials/Lensing-noise-curves*.ipynb cell 3 (the class itself is absent from the snapshot):

        nlgen = NlGenerator(shape, wcs, theory, bin_edges, lensedEqualsUnlensed=True)
        nlgen.updateNoise(beamX=, noiseTX=, noisePX=, tellminX=, tellmaxX=, pellminX=, pellmaxX=)
        ls, nls = nlgen.getNl('TT')
        ls, nls, bells, nlbb, efficiency = nlgen.getNlIterative(polCombs, kmin, kmax, tellmax, pellmin, pellmax)

    N_L^kk = L^2 (L+1)^2 A_L / 4 (legacy convention quoted in the notebooks' tracebacks); every 2-D
    quantity is computed by FFT convolution with the f64 kernels and binned with bin2D.  ``TCMB`` converts
    muK-arcmin noise to the theory's units (1 for muK^2 spectra, 2.7255e6 for dimensionless ones)."""

    def __init__(self, shape, wcs, theorySpectra, bin_edges=None, gradCut=None, TCMB=1.0, bigell=9000,
                 lensedEqualsUnlensed=False, unlensedEqualsLensed=True):
        from . import stats
        self.shape = tuple(shape[-2:])
        self.wcs = wcs
        self.geom = as_geometry(self.shape, wcs)
        self.theory = theorySpectra
        self.TCMB = TCMB
        self.gradCut = gradCut
        self.bigell = bigell
        self.use_lensed = bool(lensedEqualsUnlensed or unlensedEqualsLensed)
        self.modlmap = self.geom.modlmap()
        self.bin_edges = None if bin_edges is None else np.asarray(bin_edges, dtype=np.float64)
        self.binner = None if bin_edges is None else stats.bin2D(self.modlmap, self.bin_edges)
        self.q = None

    def updateBins(self, bin_edges):
        from . import stats
        self.bin_edges = np.asarray(bin_edges, dtype=np.float64)
        self.binner = stats.bin2D(self.modlmap, self.bin_edges)

    def updateNoise(self, beamX, noiseTX, noisePX, tellminX, tellmaxX, pellminX, pellmaxX, beamY=None, noiseTY=None,
                    noisePY=None, tellminY=None, tellmaxY=None, pellminY=None, pellmaxY=None, **kwargs):
        """White noise + Gaussian beam + ell ranges (the X and Y legs share one experiment here)."""
        ml = self.modlmap
        self.beam2d = maps.gauss_beam(ml, beamX)
        self.nT = np.full(self.shape, (noiseTX * np.pi / 180. / 60. / self.TCMB) ** 2.)
        self.nP = np.full(self.shape, (noisePX * np.pi / 180. / 60. / self.TCMB) ** 2.)
        self.tmask = maps.mask_kspace(self.shape, self.geom, lmin=tellminX, lmax=tellmaxX)
        self.pmask = maps.mask_kspace(self.shape, self.geom, lmin=pellminX, lmax=pellmaxX)
        self.q = Estimator(self.shape, self.geom, self.theory, noise2d=self.nT, beam2d=self.beam2d, kmask=self.tmask,
                           noise2d_P=self.nP, kmask_P=self.pmask, kmask_K=None, pol=True, grad_cut=self.gradCut,
                           unlensed_equals_lensed=self.use_lensed, bigell=self.bigell, dtype="f64")
        return self.nT, self.nP, self.nT, self.nP

    def _N2d(self, polComb):
        if polComb == "TT":
            return self.q.N_kappa("TT")
        self.q._setup_general(polComb)
        return self.q.N_kappa(polComb)

    def getNl(self, polComb='TT', halo=True):
        """Binned N_L^kk of one estimator: (bin centers, N_L)."""
        self.N2d = self._N2d(polComb)
        cents, nl = self.binner.bin(self.N2d)
        return cents, nl

    def getNlMV(self, polCombs):
        self.q.mv_weights(tuple(polCombs))
        cents, nl = self.binner.bin(self.q._full(self.q.Nlkk["MV"]))
        return cents, nl

    # -- flat-sky lensing B-mode power by FFT convolution:
    #    C^BB(l) = (1/Area) sum_l1 [l1.l2]^2 sin^2(a1 - a) C^EE(l1) C^pp(l2),  l2 = l - l1,  a = 2 x (mode angle)
    def lensed_bb(self, clee_h, clpp_h):
        q = self.q
        e = q.eng64
        lyd, lxd = q.ly.copy(), q.lxh.copy()
        lyd[e.ny // 2] = 0.0
        lxd[q.nxh] = 0.0
        comp = (lxd[None, :] * np.ones((e.ny, 1)), lyd[:, None] * np.ones((1, q.nxh + 1)))
        ang = q.ang_h
        cache = {}
        acc = {"1": None, "c": None, "s": None}
        for j in range(2):
            for k in range(j, 2):
                mult = 1.0 if j == k else 2.0
                dev_c128 = lambda a: _torch().as_tensor(np.ascontiguousarray(a + 0j), dtype=e.cdt, device=e.device)     # noqa: E731
                v = q._real_of(lambda: dev_c128(-comp[j] * comp[k] * clpp_h), cache, ("v", j, k))       # (i l_j)(i l_k) C^pp
                for key, trig in (("1", None), ("c", np.cos(2 * ang)), ("s", np.sin(2 * ang))):
                    base = -comp[j] * comp[k] * clee_h * (1.0 if trig is None else trig)
                    u = q._real_of(lambda: dev_c128(base), cache, ("u", j, k, key))
                    prod = e.mul_real(u, v)
                    acc[key] = e.axpby(prod, prod, mult, 0.0) if acc[key] is None else e.axpby(acc[key], prod, 1.0, mult)
        out = {}
        for key in acc:
            out[key] = e.rfft(acc[key]).cpu().numpy()[:, :q.nxh + 1].real / self.geom.pixarea
        bb = 0.5 * out["1"] - 0.5 * np.cos(2 * ang) * out["c"] - 0.5 * np.sin(2 * ang) * out["s"]
        return bb

    def getNlIterative(self, polCombs, kmin, kmax, tellmax, pellmin, pellmax, dell=20, halo=True, dTolPercentage=1.,
                       verbose=False, plot=False, max_iterations=np.inf, eff_at=60, kappa_min=0, kappa_max=np.inf):
        """Iterative EB delensing (Smith et al. 2012 style): N_L^EB -> Wiener phi -> residual lensing B power
        C^BB_res = BB[C^EE, C^pp] - BB[C^EE W^E, C^pp W^phi] -> N_L^EB ..., until the binned N_L^EB moves by less
        than dTolPercentage.  Returns (ls, N_L^MV over polCombs, bells, residual C^BB, efficiency %)."""
        from . import stats
        q = self.q
        L = q.modl_h
        with np.errstate(divide="ignore", invalid="ignore"):
            k2p = np.nan_to_num(4. / (L * (L + 1.)) ** 2)
        clkk = np.where(L <= self.bigell, self.theory.gCl("kk", L), 0.0)
        clpp = clkk * k2p
        clee = q.cl_grad["EE"]
        nee = _safe_div(q.noise["P"], q.beam ** 2)
        WE = _safe_div(clee, clee + nee) * ((L > pellmin) & (L < pellmax))
        bb_len = self.lensed_bb(clee, clpp)
        bb_res = bb_len.copy()
        old = None
        it = 0
        bells = np.arange(2, min(pellmax, 4000), dell, dtype=np.float64)
        bb_binner = stats.bin2D(self.modlmap, bells)
        while True:
            q.cl_len["BB"] = bb_res
            q._P = None
            q._gen.pop("EB", None)
            q._setup_general("EB")
            nkk = q.Nlkk["EB"]
            cents, nl_eb = self.binner.bin(q._full(nkk))
            it += 1
            if verbose:
                print("iteration", it, "N_L^EB[0..3] =", nl_eb[:3])
            if old is not None:
                change = np.nanmax(np.abs(nl_eb / old - 1.)) * 100.
                if change < dTolPercentage:
                    break
            if it >= max_iterations:
                break
            old = nl_eb
            npp = nkk * k2p
            Wp = _safe_div(clpp, clpp + npp) * ((L > max(kmin, kappa_min)) & (L < min(kmax, kappa_max)))
            bb_res = bb_len - self.lensed_bb(clee * WE, clpp * Wp)
        # MV over the requested estimators with the delensed B power
        for XY in polCombs:
            if XY != "TT":
                q._gen.pop(XY, None)
        q._mv = None
        w = q.mv_weights(tuple(polCombs))
        ls, nls = self.binner.bin(q._full(q.Nlkk["MV"]))
        bcents, nlbb = bb_binner.bin(q._full(bb_res))
        _, lbb = bb_binner.bin(q._full(bb_len))
        i = int(np.argmin(np.abs(bcents - eff_at)))
        efficiency = (1. - nlbb[i] / lbb[i]) * 100.
        return ls, nls, bcents, nlbb, efficiency
