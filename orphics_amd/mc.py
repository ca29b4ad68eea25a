"""Monte-Carlo driver: Gaussian realisations -> TT reconstruction -> bandpowers,
sharded over GPUs with the reference's task split and ONE all-reduce at the end.

This is the north-star loop of tutorials/tt_verification.ipynb cell 4 /
SURVEY.md section 3.4 for the Gaussian (N0 / mean-field) case: every realisation
is independent, so ranks never talk until ``Statistics.allreduce`` time
(stats.py:1184-1232): n (int64), sum (d,), cross (d,d) and, optionally, the
mean-field stack (Ny, kp, 2) are summed over ranks with RCCL (gloo in CPU tests).
"""
import numpy as np

from . import mpi as _mpi
from .stats import Statistics


def _torch():
    import torch
    return torch


def allreduce_tensors(tensors, comm):
    """SUM all-reduce of device (or CPU) tensors in place over a TorchComm; no-op
    for a single rank / fake comm.  One collective per tensor, issued once per run."""
    if comm is None or comm.Get_size() == 1 or not hasattr(comm, "dist"):
        return tensors
    for t in tensors:
        comm.dist.all_reduce(t, op=comm.dist.ReduceOp.SUM, group=comm.group)
    return tensors


class GaussianN0MonteCarlo(object):
    """N0 bias / mean-field Monte Carlo on Gaussian maps with total power
    C_l^TT B_l^2 + N_l, generated on the device in harmonic space
    (MapGen semantics, maps.py:1576-1587, with the Philox stream (base_seed, sim index))."""

    def __init__(self, qest, total_power_half, bin_edges, comm=None, base_seed=1234, mean_field=False, streams=1, window=None):
        """qest: lensing.Estimator; total_power_half: (Ny, Nx/2+1) host array of the
        observed-map power (C B^2 + N); bin_edges: kappa bandpower edges.
        window: (Ny, Nx) real-space apodisation (``maps.get_taper(shape, ...)[0]``) applied to every realisation before
        its transform, as the reference's analysis flow does (maps.py:1350-1361, 1873-1878): realisations are then drawn
        over the full plane and go through C2R -> x window -> R2C (``oa_mc_run_windowed``), and the mean-field stack no
        longer averages to zero.  ``window_moments`` = (mean w^2, mean w^4): bandpowers of kappa_hat (quadratic in the
        map) carry a factor mean(w^4) that the caller divides out (``debiased_mean``).
        streams > 1: this rank's simulations are split into that many contiguous blocks, each issued on its own HIP
        stream through a forked estimator handle (private plan and accumulators, summed at the end): the small
        latency-bound launches of independent realisations overlap."""
        torch = _torch()
        self.streams = max(1, int(streams))
        self.q = qest
        self.eng = qest.eng
        self.comm = comm if comm is not None else _mpi.get_world()
        self.base_seed = int(base_seed)
        self.mean_field = mean_field
        e = self.eng
        geom = qest.geom
        # unnormalised DFT of a unit-pixel-variance white map has |k|^2 = Npix; power p -> k = sqrt(p Npix^2/area) w
        amp = np.sqrt(np.asarray(total_power_half, dtype=np.float64) * float(e.npix) ** 2 / geom.area)
        self.cs = qest._hcreal(e, amp)
        self.edges = np.asarray(bin_edges, dtype=np.float64)
        self.ids = e.modl_digitize(torch.as_tensor(self.edges, device=e.device), half=True)
        self.nids = self.edges.size + 1
        self.d = self.nids - 2
        self.norm = geom.area / float(e.npix) ** 2
        # device-resident ensemble accumulators: (n, sum, cross) of the bandpower vectors and the mean-field stack
        self.acc = Statistics(comm=self.comm if hasattr(self.comm, "dist") else None, device=e.device)
        qest.bind_bins(self.ids, self.nids, self.norm)
        self.window = None
        self.window_moments = (1.0, 1.0)
        if window is not None:
            w = np.asarray(window, dtype=np.float64)
            if w.shape != (e.ny, e.nx):
                raise ValueError("window must be a (Ny, Nx) real-space array")
            self.window = e.to_real(w)
            self.window_moments = (float(np.mean(w ** 2)), float(np.mean(w ** 4)))

    def run_local(self, sims):
        """Process the given global sim indices on this rank's GPU: every contiguous block of indices is ONE
        ``oa_mc_run`` call (GRF -> TT estimator -> bandpowers -> moments [-> mean-field stack], no host work per sim)."""
        from ._lib import check
        from .engine import _ptr, _stream
        sims = [int(i) for i in sims]
        if not sims:
            return self
        q = self.q
        # (re)bind THIS driver's bins: another driver -- or a direct bind_bins call -- may have given the shared estimator
        # other edges since __init__, and the accumulators below are sized for self.d
        q.bind_bins(self.ids, self.nids, self.norm)
        e = q._bind_bins()
        n, S, C = self.acc.device_moments("n0", self.d)
        # kappa_hat vanishes outside its active region: the stack declares it, the all-reduce moves only that region
        sup = (q.kappa_rows, q.kappa_cols) if (q.kappa_cols and q.kappa_rows) else None
        mf = self.acc.device_stack("mf", (e.ny, e.kp, 2), support=sup) if self.mean_field else None
        start = prev = sims[0]
        blocks = []
        for i in sims[1:] + [None]:
            if i is None or i != prev + 1:
                blocks.append((start, prev + 1))
                start = i
            prev = i
        # (with the mean-field stack the one-stream loop measured faster -- 16.0k vs 14.1k sims/s at 4096^2 -- so the
        # lanes are used for the bandpower moments only)
        if self.window is not None and self.streams > 1 and len(sims) >= 4 * self.streams:
            # windowed realisations are a chain of ~10 launches, half of them latency-bound (draw, leg kernel, row stage,
            # divergence + binning): independent realisations on several streams overlap them with the bandwidth-bound passes
            self._run_blocks_on_streams(blocks, n, S, C, mf)
        elif self.window is not None:
            for lo, hi in blocks:
                check(e.lib.oa_mc_run_windowed(e.plan, self.base_seed, lo, hi, _ptr(self.cs), _ptr(self.window), _ptr(n), _ptr(S), _ptr(C),
                                               _ptr(mf), _stream()))
        elif self.streams > 1 and len(sims) >= 4 * self.streams and not self.mean_field:
            self._run_blocks_on_streams(blocks, n, S, C, mf)
        else:
            for lo, hi in blocks:
                check(e.lib.oa_mc_run(e.plan, self.base_seed, lo, hi, _ptr(self.cs), _ptr(n), _ptr(S), _ptr(C), _ptr(mf), _stream()))
        self.acc.note_samples("n0", len(sims))
        if self.mean_field:
            self.acc.note_stacked("mf", len(sims))
        return self

    def _run_blocks_on_streams(self, blocks, n, S, C, mf):
        """The blocks cut into ``self.streams`` pieces of (nearly) equal size; piece j runs on stream j with its own
        estimator handle and accumulators, which are added to (n, S, C, mf) in stream order at the end."""
        torch = _torch()
        from ._lib import check
        from .engine import _ptr
        K = self.streams
        if getattr(self, "_lanes", None) is None:
            lanes = []
            for j in range(K):
                qj = self.q if j == 0 else self.q.fork()
                qj.bind_bins(self.ids, self.nids, self.norm)
                own = None
                if j:     # private accumulators of this lane, allocated once and kept zero between calls
                    own = [torch.zeros_like(n), torch.zeros_like(S), torch.zeros_like(C), torch.zeros_like(mf) if mf is not None else None]
                lanes.append((qj, torch.cuda.Stream() if j else None, own))
            self._lanes = lanes
        sims = [i for lo, hi in blocks for i in range(lo, hi)]
        per = (len(sims) + K - 1) // K
        cur = torch.cuda.current_stream()
        parts = []
        for j, (qj, st, own) in enumerate(self._lanes):
            mine = sims[j * per:(j + 1) * per]
            if not mine:
                continue
            if j == 0:
                nj, Sj, Cj, mfj, stream = n, S, C, mf, cur
            else:
                nj, Sj, Cj, mfj = own
                stream = st
                stream.wait_stream(cur)
            with torch.cuda.stream(stream):
                qj.bind_bins(self.ids, self.nids, self.norm)
                ej = qj._bind_bins()
                lo = prev = mine[0]
                for i in mine[1:] + [None]:
                    if i is None or i != prev + 1:
                        if self.window is not None:
                            check(ej.lib.oa_mc_run_windowed(ej.plan, self.base_seed, lo, prev + 1, _ptr(self.cs), _ptr(self.window), _ptr(nj),
                                                            _ptr(Sj), _ptr(Cj), _ptr(mfj), stream.cuda_stream))
                        else:
                            check(ej.lib.oa_mc_run(ej.plan, self.base_seed, lo, prev + 1, _ptr(self.cs), _ptr(nj), _ptr(Sj), _ptr(Cj),
                                                   _ptr(mfj), stream.cuda_stream))
                        lo = i
                    prev = i
            if j:
                parts.append((stream, nj, Sj, Cj, mfj))
        for stream, nj, Sj, Cj, mfj in parts:
            cur.wait_stream(stream)
            n += nj; S += Sj; C += Cj
            nj.zero_(); Sj.zero_(); Cj.zero_()
            if mf is not None:
                mf += mfj
                mfj.zero_()

    def run(self, nsims):
        """Shard ``nsims`` with mpi.mpi_distribute (mpi.py:78-91), run, reduce once; returns the reduced
        :class:`Statistics` (label 'n0' = kappa auto bandpowers, stack 'mf' = interleaved (re, im) sum of the
        kappa_hat DFTs if mean_field; ``stack_sum('mf', on_device=True)`` keeps it on the GPU)."""
        comm = self.comm
        size, rank = comm.Get_size(), comm.Get_rank()
        _, tasks = _mpi.mpi_distribute(nsims, size, allow_empty=True)
        self.run_local(tasks[rank])
        self.acc.allreduce()
        return self.acc

    def debiased_mean(self):
        """mean kappa auto bandpowers of the (reduced) run divided by mean(w^4) of the window (1 without one)."""
        return self.acc.mean("n0") / self.window_moments[1]

    @property
    def centers(self):
        return (self.edges[1:] + self.edges[:-1]) / 2.


class LensedSimsMonteCarlo(object):
    """The reference's verification loop (tutorials/tt_verification.ipynb cells 4-5) as a sharded,
    device-resident driver: per realisation

        FlatLensingSims.get_sim (unlensed T,Q,U GRF -> lensing by an independent kappa GRF -> beam -> + noise; lensing.py:499-521)
        -> T, E, B transforms (FourierCalc.power2d is used only for these in the notebook)
        -> kappa_hat per estimator (qest.kappa_from_map(XY, ..., alreadyFTed=True, returnFt=True))
        -> C_b^{kappa_hat x kappa_in}, C_b^{kappa_in kappa_in} (FourierCalc.power2d + bin2D.bin)
        -> Statistics: sample = (C_b^x - C_b^in) / C_b^in per estimator (+ the two bandpower vectors themselves)

    with nothing but the (nbins,) vectors ever leaving the kernels' planes (device-side Statistics), realisations sharded
    by mpi_distribute and ONE reduce at the end.  ``stage_times=True`` brackets the stages with HIP events."""

    def __init__(self, sims, qest, bin_edges, estimators=("TT", "EB"), comm=None, base_seed=2024, lens_order=5, paired=False,
                 kappa_scale=1.0):
        """paired: every realisation is lensed TWICE, by +kappa and by -kappa, with the SAME unlensed CMB and the SAME noise;
        the estimator's odd part (kappa_hat[+] - kappa_hat[-]) / 2 keeps the terms of odd order in kappa -- the linear
        response plus O(kappa^3) -- while the Gaussian reconstruction noise (the N0 scatter that dominates the unpaired
        samples) and every even-order term cancel exactly: the normalisation is tested at the 1e-3 level with tens of
        simulations.  kappa_scale: amplitude factor on the input kappa; the O(kappa^3) part of the paired bias scales as
        kappa_scale^2, a normalisation error does not scale at all -- two values attribute a residual."""
        torch = _torch()
        self.paired, self.kappa_scale = bool(paired), float(kappa_scale)
        from . import maps
        self.sims, self.q, self.estimators = sims, qest, tuple(estimators)
        self.comm = comm if comm is not None else _mpi.get_world()
        self.base_seed, self.lens_order = int(base_seed), int(lens_order)
        e = self.eng = qest.eng
        geom = qest.geom
        self.pol = len(sims.shape) > 2 and sims.shape[-3] == 3
        if any(x != "TT" for x in self.estimators) and not self.pol:
            raise ValueError("polarised estimators need FlatLensingSims(pol=True)")
        self.fc = maps.FourierCalc(sims.shape, geom, iau=qest.iau, layout="half")
        self._rec_out = {}
        self.edges = np.asarray(bin_edges, dtype=np.float64)
        self.ids = e.modl_digitize(torch.as_tensor(self.edges, device=e.device), half=True)
        self.nids = self.edges.size + 1
        self.norm = geom.area / float(e.npix) ** 2
        # modes beyond the last bin edge only feed the overflow bin, which no sample uses: the binning kernels visit the leading
        # columns / the row band that hold every mode with ell <= edges[-1] (5 % of a 4096^2 0.5' plane for edges up to 3500)
        inside = np.asarray(geom.modlmap())[:, :e.nxh + 1] <= self.edges[-1]
        cols = np.nonzero(inside.any(axis=0))[0]
        rows = np.nonzero(inside.any(axis=1))[0]
        cw = int(cols.max()) + 1 if cols.size else 1
        rw = int(np.minimum(rows, e.ny - rows).max()) + 1 if rows.size else 1
        self._bin_region = dict(active_cols=0 if cw >= e.nxh + 1 else cw, active_rows=0 if 2 * rw - 1 >= e.ny else rw)
        self.acc = Statistics(comm=self.comm if hasattr(self.comm, "dist") else None, device=e.device)
        self.stage_ms = {}
        self.fast_sims = True         # FlatLensingSims.get_sim_teb (no transform taken twice); False: get_sim + iqu2teb, as the notebook writes it

    def _seed(self, kind, i):
        return (self.base_seed, kind, int(i))

    def run_local(self, sims_idx, stage_times=False):
        torch = _torch()
        e, q = self.eng, self.q
        ev = []

        def mark(name):
            if stage_times:
                t = torch.cuda.Event(enable_timing=True)
                t.record()
                ev.append((name, t))
        for i in sims_idx:
            mark("start")
            if self.paired:
                self._paired_sample(i)
                continue
            if self.fast_sims and hasattr(self.sims, "get_sim_teb") and not self.sims._fixed and not self.sims.iau_mismatch(q):
                # no transform taken twice (FlatLensingSims.get_sim_teb): the observed T, E, B transforms and kappa_in's directly
                teb, kin = self.sims.get_sim_teb(seed_cmb=self._seed(1, i), seed_kappa=self._seed(2, i), seed_noise=self._seed(3, i),
                                                 lens_order=self.lens_order)
                mark("get_sim")
                mark("transforms")
            else:
                parts = self.sims.get_sim(seed_cmb=self._seed(1, i), seed_kappa=self._seed(2, i), seed_noise=self._seed(3, i),
                                          lens_order=self.lens_order, return_intermediate=True)
                kappa, observed = parts[1], parts[5]
                mark("get_sim")
                teb = self.fc.iqu2teb(observed, normalize=False).t      # (3, Ny, kp) or (Ny, kp) hc planes: T, E, B
                if teb.ndim == 2:
                    teb = teb[None]
                kin = e.rfft(kappa.contiguous())
                mark("transforms")
            s_in, counts = e.bin_power(kin, kin, self.norm, self.ids, self.nids, herm=True, **self._bin_region)
            auto = s_in[1:-1] / counts[1:-1].double()
            self.acc.add("input", auto)
            f = {"T": teb[0], "E": teb[1] if self.pol else None, "B": teb[2] if self.pol else None}
            for XY in self.estimators:
                # one estimator-owned output plane per estimator, reused by every realisation: the pruned kernels write kappa_hat's
                # active region, the zero-fill of the rest (a 134 MB fill per call at 4096^2 float64) happens once
                if XY not in self._rec_out:
                    self._rec_out[XY] = q.new_output()
                out = self._rec_out[XY]
                if XY == "TT":
                    rec = q.reconstruct_tt_hc(f["T"], out=out)
                else:
                    rec = q.reconstruct_hc(XY, f[XY[0]], f[XY[1]], out=out)
                mark("qe_" + XY)
                s_x, _ = e.bin_power(rec, kin, self.norm, self.ids, self.nids, herm=True, **self._bin_region)
                cross = s_x[1:-1] / counts[1:-1].double()
                self.acc.add(XY, (cross - auto) / auto)
                self.acc.add("cross_" + XY, cross)
                mark("bandpowers")
        if stage_times and ev:
            torch.cuda.synchronize()
            tot = {}
            for (n0, t0), (n1, t1) in zip(ev[:-1], ev[1:]):
                if n1 == "start":
                    continue
                tot[n1] = tot.get(n1, 0.0) + t0.elapsed_time(t1)
            self.stage_ms = {k: v / max(1, len(sims_idx)) for k, v in tot.items()}
        return self

    def _paired_sample(self, i):
        e, q, sims = self.eng, self.q, self.sims
        unl = sims.get_unlensed(self._seed(1, i))
        kappa = sims.get_kappa(self._seed(2, i)) * self.kappa_scale
        ay, ax = sims.lenser.alpha_from_kappa(kappa)
        noise = sims.ngen.get_map(seed=self._seed(3, i))
        kin = e.rfft(kappa.contiguous())
        s_in, counts = e.bin_power(kin, kin, self.norm, self.ids, self.nids, herm=True)
        auto = s_in[1:-1] / counts[1:-1].double()
        self.acc.add("input", auto)
        recs = {XY: [] for XY in self.estimators}
        for sgn in (1.0, -1.0):
            alpha = (ay, ax) if sgn > 0 else (-ay, -ax)
            observed = sims.beam_maps(sims.lens_maps(unl, alpha, self.lens_order)) + noise
            teb = self.fc.iqu2teb(observed, normalize=False).t
            if teb.ndim == 2:
                teb = teb[None]
            f = {"T": teb[0], "E": teb[1] if self.pol else None, "B": teb[2] if self.pol else None}
            for XY in self.estimators:
                rec = q.reconstruct_tt_hc(f["T"]) if XY == "TT" else q.reconstruct_hc(XY, f[XY[0]], f[XY[1]])
                recs[XY].append(rec)
        for XY in self.estimators:
            odd = (recs[XY][0] - recs[XY][1]) * 0.5
            s_x, _ = e.bin_power(odd, kin, self.norm, self.ids, self.nids, herm=True)
            cross = s_x[1:-1] / counts[1:-1].double()
            self.acc.add(XY, (cross - auto) / auto)
            self.acc.add("cross_" + XY, cross)

    def run(self, nsims, stage_times=False):
        comm = self.comm
        size, rank = comm.Get_size(), comm.Get_rank()
        _, tasks = _mpi.mpi_distribute(nsims, size, allow_empty=True)
        self.run_local(tasks[rank], stage_times=stage_times)
        self.acc.allreduce()
        return self.acc

    def table(self):
        """per estimator: (mean bias per band, its standard error, chi^2 of the pulls, weighted mean bias +- error)"""
        out = {}
        for XY in self.estimators:
            mean = self.acc.mean(XY)
            sem = np.sqrt(self.acc.var(XY) / self.acc.count(XY))
            w = 1.0 / sem ** 2
            out[XY] = {"bias": mean, "sigma": sem, "chi2": float(np.sum((mean / sem) ** 2)), "nbands": int(mean.size),
                       "weighted_mean_bias": float(np.sum(w * mean) / np.sum(w)), "weighted_mean_sigma": float(np.sum(w) ** -0.5)}
        return out

    @property
    def centers(self):
        return (self.edges[1:] + self.edges[:-1]) / 2.
