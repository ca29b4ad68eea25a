#!/usr/bin/env python3
"""Timings of the BASELINE.json configs that are not bench.py's headline (run on the GPU box):
   config 3: 8192^2 0.5' full TT/TE/EE/EB/TB minimum-variance reconstruction on one GPU
   config 4: Monte-Carlo N0 + mean-field on 4096^2 maps (per-GPU rate of the sharded job)
   splits  : SURVEY 8f-2, SplitLensing.cross_estimator on 4 splits: the one-call device path against the reference's
             ordering of 1 + 3n + n(n-1) two-leg reconstructions through kappa_from_map
usage: python tools/config_bench.py [mv|mc|splits|all] [--n N]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from orphics_amd import cosmology, lensing, maps, mc
from orphics_amd.geometry import FlatGeometry


PREC = "f64" if "--f64" in sys.argv else "f32"      # --f64: the reference's arithmetic type


def setup(N, res, pol, prune=True):
    shape = (N, N)
    g = FlatGeometry.from_res(shape, res)
    th = cosmology.default_theory()
    ml = g.modlmap()
    beam = maps.gauss_beam(ml, 1.5)
    noise = np.full(shape, cosmology.white_noise_power(1.0))
    tmask = ((ml > 300) & (ml < 2000)).astype(np.int64)
    kmask = ((ml > 20) & (ml < 3500)).astype(np.int64)
    q = lensing.qest(shape, g, th, noise2d=noise, beam2d=beam, kmask=tmask, kmask_P=tmask, noise2d_P=2 * noise, kmask_K=kmask,
                     pol=pol, unlensed_equals_lensed=True, prune=prune, dtype=PREC)
    return shape, g, th, ml, beam, noise, q


def timeit(fn, n):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def mv(N=8192, res=0.5):
    for prune in ((True,) if "--no-dense" in sys.argv else (True, False)):
        shape, g, th, ml, beam, noise, q = setup(N, res, True, prune)
        e = q.eng
        ks = [e.grf_hc(7, i) for i in range(3)]
        out = q.new_output()                                   # estimator-owned plane: zero outside kappa's region once, not per call
        q.reconstruct_mv_hc(*ks, out=out)                      # builds every estimator's filters / normalisation
        dt = timeit(lambda: q.reconstruct_mv_hc(*ks, out=out), 10)
        plain = e.hc()                                         # any other tensor is zero-filled outside the region on every call
        dt_plain = timeit(lambda: q.reconstruct_mv_hc(*ks, out=plain), 10)
        npieces = sum(len(q._gen[x]["pieces"]) for x in ("TT", "TE", "EE", "EB", "TB") if x in q._gen)
        print("config 3: %d^2 MV (TT,TE,EE,EB,TB; %d separable leg pieces) prune=%s: %.2f ms per MV reconstruction = %.1f /s"
              " (into a caller-owned plane, zero-filled per call: %.2f ms = %.1f /s)" % (N, npieces, prune, dt * 1e3, 1 / dt, dt_plain * 1e3, 1 / dt_plain), flush=True)
        del q, ks, out
        torch.cuda.empty_cache()


def mcn0(N=4096, res=0.5, nsims=600):
    shape, g, th, ml, beam, noise, q = setup(N, res, False)
    if "--mc-batch" in sys.argv:          # realisations per launch (plan option; 1 = one by one)
        q.eng.set_option("mc_batch", int(sys.argv[sys.argv.index("--mc-batch") + 1]))
    nxh = N // 2
    tot = (th.lCl("TT", ml) * beam ** 2 + noise)[:, :nxh + 1]
    edges = np.linspace(20, 3500, 20)
    for mf, ns in (((False, 1),) if "mc1" in sys.argv else ((False, 1), (True, 1), (False, 3), (True, 3))):
        drv = mc.GaussianN0MonteCarlo(q, tot, edges, mean_field=mf, streams=ns)
        drv.run_local(range(24))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        drv.run_local(range(24, 24 + nsims))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / nsims
        print("config 4: %d^2 MC N0%s, %d stream(s): %.3f ms per simulated realisation = %.0f sims/s per GPU (measured on ONE GPU; the 8-GPU "
              "job was not run by the builder: 125 sims per rank would take %.3f s of compute + one all-reduce, an extrapolation)"
              % (N, " + mean-field stack" if mf else "", ns, dt * 1e3, 1 / dt, 125 * dt), flush=True)
    # the reference's analysis flow: every realisation multiplied by its apodisation taper before the transform
    # (maps.get_taper; oa_mc_run_windowed: full-plane draw -> C2R -> x taper -> R2C on the active columns -> QE -> moments + stack)
    taper, w2 = maps.get_taper(shape, g)
    for mf in (False, True):
        drv = mc.GaussianN0MonteCarlo(q, tot, edges, mean_field=mf, window=taper)
        drv.run_local(range(12))
        torch.cuda.synchronize()
        nw = max(60, nsims // 4)
        t0 = time.perf_counter()
        drv.run_local(range(12, 12 + nw))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / nw
        print("config 4 WINDOWED (12 %% cosine taper, mean w^2 = %.3f): %d^2 MC N0%s: %.3f ms per realisation = %.0f sims/s per GPU"
              % (w2, N, " + mean-field stack" if mf else "", dt * 1e3, 1 / dt), flush=True)


def splits(N=8192, res=0.5, n=4):
    from orphics_amd.stats import HalfPlane
    shape, g, th, ml, beam, noise, q = setup(N, res, False)
    e = q.eng
    half = HalfPlane(torch.stack([e.grf_hc(11, i) for i in range(n)]), e)
    sl = lensing.SplitLensing(shape, g, q, "TT")
    dt = timeit(lambda: sl.cross_estimator(half), 10)

    own = q.new_output()                       # estimator-owned plane: no zero-fill per reconstruction on either path

    def reference_order():                     # the calls the reference's loop makes (lensing.py:980-1003), on device planes
        m = [half.t[i] for i in range(n)]
        s = half.t.mean(dim=0)
        q.reconstruct_tt_hc(s, s, out=own)
        for i in range(n):
            q.reconstruct_tt_hc(m[i], s, out=own); q.reconstruct_tt_hc(s, m[i], out=own); q.reconstruct_tt_hc(m[i], m[i], out=own)
            for j in range(i + 1, n):
                q.reconstruct_tt_hc(m[i], m[j], out=own); q.reconstruct_tt_hc(m[j], m[i], out=own)
    dt_ref = timeit(reference_order, 5)
    print("splits: %d^2 SplitLensing.cross_estimator on %d splits: %.2f ms per estimate (one oa_qe_tt_splits call + combination) = %.0f /s;"
          " the reference's %d reconstructions one by one (without its power / combination arithmetic): %.2f ms"
          % (N, n, dt * 1e3, 1 / dt, 1 + 3 * n + n * (n - 1), dt_ref * 1e3), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("mv", "all"):
        mv()
    if what in ("mc", "mc1", "all"):          # mc1: one stream without the mean-field stack only (tools/trace_mc.sh)
        mcn0()
    if what in ("splits", "all"):
        splits()
