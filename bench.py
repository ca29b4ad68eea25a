#!/usr/bin/env python3
"""Benchmark of the hot path: full TT quadratic-estimator kappa reconstructions
per second on N^2 0.5' maps (BASELINE.json metric; default N = 8192).

One "step" = one reconstruction from a real-space map resident in HBM:
  row R2C -> forward column pass 1 -> [forward column pass 2 + leg filters + inverse column pass 1] -> inverse
  column pass 2 (3 planes, one launch) -> [fused row stage: 3 C2R, 2 products, 2 R2C in LDS] -> forward column
  pass 1 (2 planes, one launch) -> [forward column pass 2 + divergence * A_L] -> [|kappa_hat|^2 + radial
  bandpowers] -> [bin means + moment accumulation]                                ([...] = one fused kernel).
Multi-GPU: independent realisations per rank (weak scaling, no data-path
collective) + ONE RCCL all-reduce of the bandpower moments at the end.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints one JSON line (rank 0) with `roofline` (dominant kernel, live HIP-event
timing on the launch stream) and `cpu_baseline` (NumPy oracle on host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes on this driver)
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s achievable)


def build_pipeline(N, res_arcmin, prec, torch, prune=True):
    from orphics_amd import cosmology, lensing, maps
    from orphics_amd.geometry import FlatGeometry
    shape = (N, N)
    geom = FlatGeometry.from_res(shape, res_arcmin)
    theory = cosmology.default_theory()
    nxh = N // 2
    ly, lx = geom.laxes()
    ml_h = np.sqrt(ly[:, None] ** 2 + lx[None, :nxh + 1] ** 2)
    # half-plane inputs are expanded only where the public constructor wants full planes;
    # here everything is even-symmetric, so build full planes by mirroring cheaply
    def full(a_h):
        out = np.empty(shape, dtype=a_h.dtype)
        out[:, :nxh + 1] = a_h
        idx = (-np.arange(N)) % N
        out[:, nxh + 1:] = a_h[idx][:, 1:nxh][:, ::-1]
        return out
    beam_h = maps.gauss_beam(ml_h, 1.5)
    noise_h = np.full(ml_h.shape, cosmology.white_noise_power(1.0))
    tmask_h = ((ml_h > 300) & (ml_h < 2000)).astype(np.int64)
    kmask_h = ((ml_h > 20) & (ml_h < 3500)).astype(np.int64)
    qkw = dict(noise2d=full(noise_h), beam2d=full(beam_h), kmask=full(tmask_h), kmask_K=full(kmask_h),
               unlensed_equals_lensed=True, dtype=prec)
    q = lensing.qest(shape, geom, theory, prune=prune, **qkw)
    eng = q.eng
    # synthetic observed maps: GRF with C_l^TT B^2 + N
    cl_h = theory.lCl("TT", ml_h)
    cs = np.sqrt((cl_h * beam_h ** 2 + noise_h) * (N * N) / geom.area)
    cs_d = eng.hcreal()
    cs_d[:, :nxh + 1] = torch.as_tensor(cs, dtype=eng.rdt, device=eng.device)
    edges = np.linspace(20, 3500, 20)
    ed = torch.as_tensor(edges, device=eng.device)
    ids = eng.modl_digitize(ed, half=True)
    return dict(q=q, qkw=qkw, eng=eng, geom=geom, cs=cs_d, ids=ids, nids=len(edges) + 1, edges=edges, theory=theory,
                beam_h=beam_h, noise_h=noise_h, tmask_h=tmask_h, kmask_h=kmask_h, cl_h=cl_h)


def timed_steps(torch, step, nsteps, nwarm=10):
    for i in range(nwarm):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(nsteps):
        step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / nsteps


def bandlimited_leg(P, args, torch, tmaps, ref_p1d):
    """Same job (R2C of the full-resolution map -> kappa_hat -> 19 bandpowers) with the reconstruction on the
    smallest grid that holds the band-limited legs and their products exactly (lensing.BandlimitedEstimator)."""
    from orphics_amd import lensing
    N = args.n
    bl = lensing.BandlimitedEstimator((N, N), P["geom"], P["theory"], **P["qkw"])
    es = bl.q.eng
    ids = es.modl_digitize(torch.as_tensor(P["edges"], device=es.device), half=True)
    nrm = bl.gsmall.area / float(bl.n ** 2) ** 2
    kk = es.hc()
    _, counts = es.bin_power(kk, kk, nrm, ids, P["nids"], herm=True)
    res = {}

    def step(i):
        bl.reconstruct_tt_from_map(tmaps[i & 1], out=kk)
        res["sums"], _ = es.bin_power(kk, kk, nrm, ids, P["nids"], herm=True, active_cols=bl.q.kappa_cols, active_rows=bl.q.kappa_rows)
    dt = timed_steps(torch, step, args.steps)
    step(0)
    p1d = res["sums"][1:-1] / counts[1:-1]
    return {"reconstructions_per_s": 1.0 / dt, "internal_grid": bl.n,
            "max_rel_bandpower_diff_vs_headline_path": float((p1d / ref_p1d - 1).abs().max().item()),
            "note": "opt-in lensing.BandlimitedEstimator: input R2C at full resolution (active columns only), estimator "
                    "on the coarse grid; exact for band-limited filters (coarse Nyquist > ell_max_X + ell_max_Y)"}


def dense_leg(P, args, torch, tmaps, ref_p1d, norm):
    """The same job with prune=False: every plane processed over all nx/2+1 columns (what a filter without a
    band limit costs).  Reported beside the headline for transparency."""
    from orphics_amd import lensing
    q = lensing.qest((args.n, args.n), P["geom"], P["theory"], prune=False, **P["qkw"])
    eng = q.eng
    ns = max(1, args.streams)
    qs = [q] + [q.fork() for _ in range(ns - 1)]
    streams = [torch.cuda.Stream() for _ in range(ns)]
    kks = [e.eng.hc() for e in qs]
    res = {}

    def step(i):
        j = i % ns
        with torch.cuda.stream(streams[j]):
            e = qs[j].eng
            qs[j].reconstruct_tt_from_map(tmaps[i & 1], out=kks[j])
            res[i & 1] = e.bin_power(kks[j], kks[j], norm, P["ids"], P["nids"], herm=True)
    dt = timed_steps(torch, step, args.steps)
    step(0)
    torch.cuda.synchronize()
    sums, counts = res[0]
    p1d = sums[1:-1] / counts[1:-1]
    A = 4 * args.n * args.n if args.prec == "f32" else 8 * args.n * args.n
    return {"reconstructions_per_s": 1.0 / dt, "streams_per_gpu": ns,
            "max_rel_bandpower_diff_vs_headline_path": float((p1d / ref_p1d - 1).abs().max().item()),
            "pipeline_achieved_GBs_on_survey_37.25A": 37.25 * A / dt / 1e9,
            "pipeline_frac_of_hbm_peak": 37.25 * A / dt / 1e9 / HBM_PEAK_GBS,
            "note": "prune=False: all nx/2+1 columns of every plane are transformed (filters without a band limit)"}


def time_kernel(torch, fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3  # seconds per launch


def cpu_baseline(N_gpu, res_arcmin, budget_n=None):
    """NumPy/SciPy oracle (float64, full-plane C2C like the reference) on a bounded sample."""
    from oracle import maps_oracle as mo
    from oracle import qe_oracle as qo
    from oracle import stats_oracle as so
    cores = os.cpu_count() or 1
    if budget_n is None:
        budget_n = 8192 if cores >= 64 else 4096     # keep the sample to ~10-30 s of CPU work
    Ns = min(N_gpu, budget_n)
    mo.set_workers(cores)
    res = res_arcmin * np.pi / 180. / 60.
    shape = (Ns, Ns)
    rng = np.random.default_rng(0)
    ml = mo.modlmap(shape, res, -res)
    mask = ((ml > 300) & (ml < 2000)).astype(np.float64)
    Wg = mask / (1.0 + ml)
    Wh = mask / (1.0 + ml)
    Fn = ((ml > 20) & (ml < 3500)) * 1e-3
    q = qo.QEOracleTT.for_timing(shape, res, -res, Wg, Wh, Fn)
    fc = mo.FourierCalc(shape, res, -res)
    binner = so.bin2D(ml, np.linspace(20, 3500, 20))
    tmap = rng.standard_normal(shape)
    t0 = time.perf_counter()
    kT = fc.fft(tmap)
    kk = q.kappa_ft(kT)
    p2d = fc.f2power(kk, kk)
    binner.bin(p2d)
    dt = time.perf_counter() - t0
    # scale N^2 log2(N^2) to the GPU workload size
    scale = (N_gpu / Ns) ** 2 * (np.log2(float(N_gpu)) / np.log2(float(Ns)))
    return {"value": 1.0 / (dt * scale), "unit": "reconstructions/s", "cores": cores, "kind": "port",
            "sample": "one %dx%d float64 full-plane reconstruction (%.1f s), scaled x%.2f (N^2 log N) to %dx%d"
                      % (Ns, Ns, dt, scale, N_gpu, N_gpu)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n", type=int, default=8192, help="map side (default 8192, the metric's size)")
    ap.add_argument("--res", type=float, default=0.5)
    ap.add_argument("--prec", default="f32", choices=["f32", "f64"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-prune", action="store_true",
                    help="process all nx/2+1 columns of every plane even where the band-limited filters vanish")
    ap.add_argument("--trace-steps", action="store_true", help="stderr: throughput per 20 timed steps (diagnostic)")
    ap.add_argument("--preroll", type=float, default=1.5, help="seconds of untimed load before the warm-up steps (clock ramp)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the two side measurements reported under 'extra' (never the headline value): the dense "
                         "pipeline (prune=False) and the opt-in coarse-grid lensing.BandlimitedEstimator")
    ap.add_argument("--streams", type=int, default=2, help="HIP streams: independent realisations are issued round-robin "
                    "on this many streams (each with its own plan/workspace) so latency-bound and bandwidth-bound kernels overlap")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    backend = os.environ.get("OA_BENCH_BACKEND", "nccl")      # "gloo": rehearse the N>1 path on a box with fewer GPUs
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    N = args.n
    P = build_pipeline(N, args.res, args.prec, torch, prune=not args.no_prune)
    q, eng = P["q"], P["eng"]
    nids = P["nids"]
    d = nids - 2
    norm = P["geom"].area / float(N * N) ** 2

    # two resident input maps (distinct realisations per rank)
    tmaps = [eng.irfft(eng.grf_hc(1234 + rank, i, P["cs"]), scale=1.0 / np.sqrt(eng.npix)) for i in range(2)]
    ns = max(1, args.streams)
    qs = [q] + [q.fork() for _ in range(ns - 1)]              # shared filters, private plan + work buffers
    # every realisation stream is a side stream (measured ~3 % better than pairing the default stream with one)
    streams = [torch.cuda.Stream() for _ in range(ns)]
    kTs, kks = [e.eng.hc() for e in qs], [e.eng.hc() for e in qs]
    kT, kk = kTs[0], kks[0]
    p2d = eng.hcreal()
    mom_n = [torch.zeros(1, dtype=torch.int64, device=eng.device) for _ in range(ns)]
    mom_S = [torch.zeros(d, dtype=torch.float64, device=eng.device) for _ in range(ns)]
    mom_C = [torch.zeros(d, d, dtype=torch.float64, device=eng.device) for _ in range(ns)]
    from orphics_amd.engine import _ptr, _stream
    from orphics_amd._lib import check

    wl, wk = q.leg_cols, q.kappa_cols
    # mode counts per bin do not depend on the data: taken once over the whole plane
    _, counts = eng.bin_power(kT, kT, norm, P["ids"], nids, herm=True)
    torch.cuda.synchronize()                 # `counts` is read from every stream below

    def step(i):
        j = i % ns
        with torch.cuda.stream(streams[j]):
            e = qs[j].eng
            # map -> kappa_hat: columns / rows beyond the filters' support are neither produced nor read (exact: the
            # masks zero them); the map's transform is consumed inside the fused leg kernel
            qs[j].reconstruct_tt_from_map(tmaps[i & 1], out=kks[j])
            sums, _ = e.bin_power(kks[j], kks[j], norm, P["ids"], nids, herm=True, active_cols=wk, active_rows=q.kappa_rows)   # |kappa_hat|^2 binned in one kernel
            # bin means (bin2D.bin) + ensemble moments (Statistics.add_to_stats) in one small kernel
            check(e.lib.oa_moments_add_binned(_ptr(sums[1:]), _ptr(counts[1:]), d, _ptr(mom_n[j]), _ptr(mom_S[j]), _ptr(mom_C[j]), _stream()))

    # pre-roll: a fresh box idles at ~550 MHz sclk and needs a few hundred ms of load to reach its sustained
    # clocks; W warm-up steps alone (~10 ms) would leave the ramp inside the timed region.  Untimed, uncounted.
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.preroll:
        for i in range(8):
            step(i)
        torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    # rehearse the end-of-job reduction once (first use of a torch op / of the RCCL communicator loads code
    # objects and opens connections: tens of ms that belong to start-up, not to the K timed steps)
    wn, wS, wC = sum(mom_n), sum(mom_S), sum(mom_C)
    if world > 1:
        dist.all_reduce(wn); dist.all_reduce(wS); dist.all_reduce(wC)
    torch.cuda.synchronize()
    del wn, wS, wC
    for j in range(ns):                      # the timed region counts only its own realisations
        mom_n[j].zero_(); mom_S[j].zero_(); mom_C[j].zero_()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs = []
    for i in range(args.steps):
        step(i)
        if args.trace_steps:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(streams[i % ns])
            evs.append(ev)
    t_issue = time.perf_counter() - t0          # host time to enqueue the K steps (diagnostic: must stay < elapsed)
    torch.cuda.synchronize()
    if args.trace_steps and rank == 0:
        ts = [evs[0].elapsed_time(e) for e in evs]
        sys.stderr.write("event span first->last step end: %.2f ms; wall to issue %.2f ms\n" % (ts[-1], t_issue * 1e3))
        for a in range(0, len(ts) - 20, 20):
            sys.stderr.write("steps %4d-%4d: %.1f recon/s\n" % (a, a + 20, 20.0 / max(ts[a + 20] - ts[a], 1e-9) * 1e3))
    mom_n, mom_S, mom_C = sum(mom_n), sum(mom_S), sum(mom_C)     # per-stream accumulators
    if world > 1:
        # the ensemble reduce of Statistics.allreduce (stats.py:1209-1230): n, sum, cross
        dist.all_reduce(mom_n)
        dist.all_reduce(mom_S)
        dist.all_reduce(mom_C)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=eng.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total = int(mom_n.item())
    assert total == args.steps * max(world, 1), "moment counter %d != steps x ranks" % total

    if rank == 0:
        es = 4 if args.prec == "f32" else 8
        A = es * N * N                        # one real plane
        W = N // 2 + 1                        # hc columns
        Ah = 2 * es * N * W                   # one half-complex plane (valid columns)
        fl = (wl or W) / float(W)             # active fraction of the leg planes / of the input transform
        fk = (wk or W) / float(W)             # active fraction of the product / kappa planes
        # ---- live per-kernel timing (HIP events on the launch stream) ----
        # per launch: (launcher, SURVEY-8d algorithmic bytes of the stages it covers -- 2-D FFT = row stage 2A +
        # column stage 2A, my two column passes carry A each; plane terms scale with the active-column fraction
        # of the plane they move --, bytes the kernel itself must move once (inputs + outputs), launches/recon)
        s1, s2, s3, s4, s5 = eng.hc(), eng.hc(), eng.hc(), eng.hc(), eng.hc()
        r1 = eng.real()
        FG, FH, Fn = q._F["TT"]
        lib = eng.lib

        def map_legs():     # row R2C + forward column pass 1 + fused (forward pass 2, legs, inverse pass 1) + 3-plane pass 2
            eng.qe_map_legs_cols(tmaps[0], FG, FH, out=(s1, s2, s3), width=wl, rband=q.leg_rows)

        kern = {
            "row_fft_kernel<R2C>": (lambda: eng.fft_pass(0, r1, s1, wl), (1 + fl) * A, A + fl * Ah, 1),
            "col_fft_kernel<pass1,legs-width>": (lambda: eng.fft_pass(1, s1, s2, wl), fl * A, 2 * fl * Ah, 1),
            # whole C-ABI call oa_qe_map_legs_cols = the two entries above + col_fwdlegs_kernel (second half of the
            # forward column stage, filter multiply, first half of the 3 inverse column stages) + ONE 3-plane launch
            # of their second half
            "map_legs_cols (whole call)": (map_legs, (1 + fl) * A + 2 * fl * A + (4 * fl + 6 * fl) * A,
                                           A + fl * Ah + 2 * fl * Ah + fl * (Ah + 2 * Ah / 2 + 3 * Ah) + 6 * fl * Ah, 1),
            "row_qe_kernel": (lambda: eng.qe_rows(s1, s2, s3, s4, s5, win=wl, wout=wk), (10 + 3 * fl + 2 * fk) * A, (3 * fl + 2 * fk) * Ah, 1),
            # ONE 2-plane launch of the first half of the 2 forward column stages + col_div (second half + divergence)
            "cols_div = col_fft_kernel<pass1 x2 planes> + col_div_kernel": (lambda: eng.qe_cols_div(s4, s5, Fn, out=kk, width=wk, rband=q.kappa_rows),
                                                                           (4 * fk + 3 * fk) * A, 4 * fk * Ah + fk * (2 * Ah + Ah / 2 + Ah), 1),
            "bin_kernel<power>": (lambda: eng.bin_power(kk, kk, norm, P["ids"], nids, herm=True, active_cols=wk, active_rows=q.kappa_rows), 2.75 * fk * A, 1.5 * fk * Ah, 1),
        }
        per, share = {}, {}
        for name, (fn, alg, actual, count) in kern.items():
            dt = time_kernel(torch, fn)
            share[name] = dt * 1e3 * count
            per[name] = {"avg_ms": dt * 1e3, "launches_per_recon": count, "algorithmic_GB": alg / 1e9,
                         "hbm_min_GB": actual / 1e9, "achieved_GBs": alg / dt / 1e9, "achieved_actual_GBs": actual / dt / 1e9}
        whole = "map_legs_cols (whole call)"
        t_fl = max(per[whole]["avg_ms"] - per["row_fft_kernel<R2C>"]["avg_ms"] - per["col_fft_kernel<pass1,legs-width>"]["avg_ms"], 1e-6)
        name_fl = "fwdlegs_cols = col_fwdlegs_kernel + col_fft_kernel<pass2 x3 planes>"
        a_fl, m_fl = (fl + 4 * fl + 6 * fl) * A, fl * (Ah + 2 * Ah / 2 + 3 * Ah) + 6 * fl * Ah
        per[name_fl] = {"avg_ms": t_fl, "launches_per_recon": 1, "algorithmic_GB": a_fl / 1e9, "hbm_min_GB": m_fl / 1e9,
                        "achieved_GBs": a_fl / t_fl / 1e6, "achieved_actual_GBs": m_fl / t_fl / 1e6,
                        "derived": "whole call minus its row and pass-1 launches"}
        share[name_fl] = t_fl
        del share[whole]
        # the fused row stage is arithmetic, not bandwidth, bound: 5 packed N/2-point complex transforms per row
        # (5 L log2 L flop each) against the f32 vector peak (MI355X_MICROARCH.md: 157.3 TFLOP/s counts every lane
        # as an FMA; an FFT is ~2/3 additions, so ~50 % of it is the practical ceiling of butterfly code)
        Lrow = N // 2
        qe_flop = N * 5 * 5.0 * Lrow * np.log2(Lrow)
        per["row_qe_kernel"]["fft_GFLOP_per_launch"] = qe_flop / 1e9
        per["row_qe_kernel"]["achieved_TFLOPs"] = qe_flop / (per["row_qe_kernel"]["avg_ms"] * 1e-3) / 1e12
        per["row_qe_kernel"]["frac_of_f32_vector_peak_157.3"] = per["row_qe_kernel"]["achieved_TFLOPs"] / 157.3
        dom = max(share, key=share.get)
        d_alg, d_act, d_t = per[dom]["algorithmic_GB"] * 1e9, per[dom]["hbm_min_GB"] * 1e9, per[dom]["avg_ms"]
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_r01.json")
        if os.path.exists(tpath) and N == 8192 and args.prec == "f32" and not args.no_prune:
            try:
                traffic = json.load(open(tpath)).get(dom)
            except Exception:
                traffic = None
        # canonical stage model of SURVEY 8d restricted to the active columns of each plane it moves:
        # FFT(T) (1+3fl) + filter multiply 4fl + 3 inverse FFTs 3(3fl+1) + products 5 + 2 forward FFTs 2(1+3fk)
        # + divergence 3fk + power/bin 1.25fk  (fl = fk = 1 gives the survey's 37.25 A)
        alg_recon = (11 + 16 * fl + 10.25 * fk) * A
        rate = total / elapsed / max(world, 1)
        roofline = {"bound": "hbm", "kernel": dom, "achieved": d_alg / d_t / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": d_alg / d_t / 1e6 / HBM_PEAK_GBS, "traffic": traffic,
                    "algorithmic_bytes_per_launch": d_alg, "hbm_min_bytes_per_launch": d_act,
                    "achieved_on_hbm_min_bytes": d_act / d_t / 1e6, "frac_on_hbm_min_bytes": d_act / d_t / 1e6 / HBM_PEAK_GBS,
                    "note": "achieved = SURVEY 8d algorithmic bytes of the stages the (fused) kernel covers, restricted to the "
                            "active columns, / live HIP-event duration; a fused kernel keeps most of those bytes in LDS/registers, "
                            "so achieved can exceed the HBM peak -- *_hbm_min_* = bytes the kernel itself must move",
                    "active_columns": {"legs": wl or W, "kappa": wk or W, "of": W},
                    "share_of_recon_ms": share, "per_kernel": per,
                    "pipeline": {"algorithmic_bytes_per_recon": alg_recon, "achieved_GBs": alg_recon * rate / 1e9,
                                 "frac": alg_recon * rate / 1e9 / HBM_PEAK_GBS,
                                 "survey_unpruned_bytes_per_recon": 37.25 * A,
                                 "survey_unpruned_equivalent_GBs": 37.25 * A * rate / 1e9}}
        out = {
            "metric": "QE kappa reconstructions/sec on %d^2 maps" % N,
            "value": total / elapsed, "unit": "reconstructions/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "host_issue_ms_per_step": t_issue / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.prec, "data": "synthetic",
            "config": {"workload": "TT quadratic estimator (lensing.Estimator) on %dx%d %.2f-arcmin flat-sky GRF maps, "
                                   "incl. R2C of the input map and 19-bin kappa auto-bandpowers; T filter ell in (300,2000), "
                                   "1.5' beam, 1 uK' noise" % (N, N, args.res),
                       "map_side": N, "res_arcmin": args.res, "estimator": "TT", "nbins": d,
                       "streams_per_gpu": ns,
                       "parallelism": "independent realisations per GPU + 1 all-reduce of bandpower moments"},
            "roofline": roofline,
        }
        if world == 1 and not args.no_extras and not args.no_prune:
            # bandpowers of map 0 through the headline path: the yardstick for the two side measurements
            e0 = qs[0].eng
            qs[0].reconstruct_tt_from_map(tmaps[0], out=kks[0])
            s0, _ = e0.bin_power(kks[0], kks[0], norm, P["ids"], nids, herm=True, active_cols=wk, active_rows=q.kappa_rows)
            ref_p1d = s0[1:-1] / counts[1:-1]
            out["extra"] = {"dense": dense_leg(P, args, torch, tmaps, ref_p1d, norm),
                            "bandlimited": bandlimited_leg(P, args, torch, tmaps, ref_p1d)}
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(N, args.res)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
