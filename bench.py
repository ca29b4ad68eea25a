#!/usr/bin/env python3
"""Benchmark of the hot path: full TT quadratic-estimator kappa reconstructions
per second on N^2 0.5' maps (BASELINE.json metric; default N = 8192).

One "step" = one pass of the hot path over one BATCH of --batch (default 64) independent real-space maps resident in HBM
(the shard of realisations a GPU is handed in the Monte-Carlo job); value = reconstructions/s =
K x batch x ranks / elapsed.  (The driver times K = 20 steps: 20 single reconstructions would be a 2.6 ms timed region,
a fifth of which is the clock / pipeline ramp after the synchronising barrier -- 7.6 k/s measured against 9.1 k/s
sustained.)  Each reconstruction of the batch:
  [row R2C + first radix-4 butterfly of the column transform] -> [forward columns + leg filters + inverse columns on the
  column grid: one single-pass kernel] -> [fused row stage: 3 C2R, 2 products, 2 R2C in LDS] -> [single-pass forward
  columns + divergence * A_L] -> [|kappa_hat|^2 + radial bandpowers] -> [bin means + moment accumulation]
                                                                                  ([...] = one fused kernel).
Multi-GPU: independent realisations per rank (weak scaling, no data-path
collective) + ONE RCCL all-reduce of the bandpower moments at the end.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no torchrun environment the script starts N ranks itself (a child
``python -m torch.distributed.run`` started BEFORE this process touches the GPU);
under torchrun it reads RANK / LOCAL_RANK / WORLD_SIZE and checks them against --gpus.

Prints one JSON line (rank 0).  The HEADLINE (`value`, `dtype`, `roofline`) is measured in float64 / complex128 -- the
reference's arithmetic (maps.py:1613) --; the same job through the float32 kernels is the second, equally complete block
`f32` (its own timed region, roofline, per-kernel table and PMC traffic):
  value / ms_per_step : whole-job reconstructions/s over the K timed steps = K batches (max over ranks); ms per batch
  roofline            : the dominant kernel against the roof that binds it -- "valu" (f32 vector peak, on the
                        arithmetic it executes) for the fused row stage, "hbm" (on the bytes it moves) otherwise;
                        `frac` <= 1 by construction.  `hbm` holds the memory side: the dense pipeline against
                        SURVEY 8d's 37.25 A bytes and every bandwidth-bound kernel on its own bytes.
  extra               : side legs of the same job (never the headline): dense (prune=False), bandlimited,
                        f64 (the reference's arithmetic type), wideband (T filter ell < 6000)
  cpu_baseline        : NumPy/SciPy oracle on the host cores (workers = 1 and = nproc, median of 5 after warm-up)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes on this driver)
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s achievable)
VALU_PEAK_TFLOPS = 157.3   # f32 vector peak (MI355X_MICROARCH.md chip table)
PROFILE_TAG = "r05"        # profiles/traffic_<tag>_<prec>[_dense|_fullrows].json (tools/collect_profiles.sh)


# --------------------------------------------------------------------------------------------------------------
# launch plumbing
# --------------------------------------------------------------------------------------------------------------
def resolve_world(gpus, env):
    """(world, rank, local_rank, spawn): ``spawn`` = this process must start the ranks itself.
    Raises SystemExit when the torchrun environment contradicts --gpus."""
    if "WORLD_SIZE" in env:
        world = int(env["WORLD_SIZE"])
        if world != gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU: python -m "
                             "torch.distributed.run --nproc-per-node %d bench.py --gpus %d ...)" % (gpus, world, gpus, gpus))
        return world, int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0")), False
    if gpus > 1:
        return gpus, 0, 0, True
    return 1, 0, 0, False


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_gpu_count():
    """Number of GPUs a child process will see, found WITHOUT initialising HIP in this process (the ranks are started
    before the parent touches the GPU; torch.cuda.device_count() may call hipGetDeviceCount): the *_VISIBLE_DEVICES
    environment first, else the KFD topology (nodes with SIMDs are GPUs); None = unknown (each rank validates itself)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    root = "/sys/class/kfd/kfd/topology/nodes"
    if not os.path.isdir(root):
        return 0                      # no KFD driver: no AMD GPU on this host
    try:
        n = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as f:
                for line in f:
                    if line.startswith("simd_count") and int(line.split()[1]) > 0:
                        n += 1
        return n
    except Exception:
        return None


def spawn_ranks(gpus, argv):
    """Start ``gpus`` ranks as CHILD processes, one per GPU, with the torchrun environment (RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR, MASTER_PORT).  This parent imports neither torch nor the HIP library, makes no GPU call
    and never execs; it returns non-zero if any rank fails (the others are then terminated)."""
    backend = os.environ.get("OA_BENCH_BACKEND", "nccl")
    ndev = visible_gpu_count()
    if backend == "nccl" and ndev is not None and ndev < gpus:
        raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible" % (gpus, ndev))
    port = str(_free_port())
    procs = []
    for r in range(gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(gpus), LOCAL_WORLD_SIZE=str(gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for other in pending:          # one rank died: the rest would hang in the next collective
                    other.terminate()
        time.sleep(0.05)
    return rc


# --------------------------------------------------------------------------------------------------------------
# pipeline construction
# --------------------------------------------------------------------------------------------------------------
def build_pipeline(N, res_arcmin, prec, torch, prune=True, tlmax=2000.0, klmax=3500.0, row_grid="auto"):
    from orphics_amd import cosmology, lensing, maps
    from orphics_amd.geometry import FlatGeometry
    shape = (N, N)
    geom = FlatGeometry.from_res(shape, res_arcmin)
    theory = cosmology.default_theory()
    nxh = N // 2
    ly, lx = geom.laxes()
    ml_h = np.sqrt(ly[:, None] ** 2 + lx[None, :nxh + 1] ** 2)

    # the public constructor takes full planes; everything here is even-symmetric: mirror the half planes
    def full(a_h):
        out = np.empty(shape, dtype=a_h.dtype)
        out[:, :nxh + 1] = a_h
        idx = (-np.arange(N)) % N
        out[:, nxh + 1:] = a_h[idx][:, 1:nxh][:, ::-1]
        return out
    beam_h = maps.gauss_beam(ml_h, 1.5)
    noise_h = np.full(ml_h.shape, cosmology.white_noise_power(1.0))
    tmask_h = ((ml_h > 300) & (ml_h < tlmax)).astype(np.int64)
    kmask_h = ((ml_h > 20) & (ml_h < klmax)).astype(np.int64)
    qkw = dict(noise2d=full(noise_h), beam2d=full(beam_h), kmask=full(tmask_h), kmask_K=full(kmask_h),
               unlensed_equals_lensed=True, dtype=prec)
    q = lensing.qest(shape, geom, theory, prune=prune, row_grid=row_grid, **qkw)
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
                tlmax=tlmax, prec=prec)


def make_maps(P, torch, seed, n=2):
    eng = P["eng"]
    return [eng.irfft(eng.grf_hc(seed, i, P["cs"]), scale=1.0 / np.sqrt(eng.npix)) for i in range(n)]


class Runner(object):
    """`ns` estimator handles on `ns` HIP streams (shared filters, private plan + work buffers), device-side
    bandpower moments per stream: step(i) = map -> kappa_hat -> 19 bandpowers -> (n, S, C) accumulation."""

    def __init__(self, P, torch, tmaps, ns, pair=True):
        from orphics_amd.engine import _ptr, _stream
        from orphics_amd._lib import check
        self.P, self.torch, self.tmaps, self.ns = P, torch, tmaps, max(1, ns)
        self.pair = bool(pair)        # two realisations per C-ABI call (oa_qe_tt_moments2): they share every launch behind the row R2C
        q, eng = P["q"], P["eng"]
        self.q, self.eng = q, eng
        self.d = P["nids"] - 2
        N = eng.ny
        self.norm = P["geom"].area / float(N * N) ** 2
        self.qs = [q] + [q.fork() for _ in range(self.ns - 1)]
        self.streams = [torch.cuda.Stream() for _ in range(self.ns)]
        self.kks = [e.new_output() for e in self.qs]
        d = self.d
        self.mom_n = [torch.zeros(1, dtype=torch.int64, device=eng.device) for _ in range(self.ns)]
        self.mom_S = [torch.zeros(d, dtype=torch.float64, device=eng.device) for _ in range(self.ns)]
        self.mom_C = [torch.zeros(d, d, dtype=torch.float64, device=eng.device) for _ in range(self.ns)]
        # radial bins bound to every handle's plan: one C-ABI call per step (oa_qe_tt_moments)
        for e in self.qs:
            e.bind_bins(P["ids"], P["nids"], self.norm)
        self.counts = q.bin_counts()            # data-independent mode counts per bin (whole plane)
        torch.cuda.synchronize()
        self._ptr, self._stream, self._check = _ptr, _stream, check

    def step(self, i):
        """map -> kappa_hat -> 19 bandpowers -> moments: ONE C-ABI call (oa_qe_tt_moments) on this step's stream"""
        j = i % self.ns
        with self.torch.cuda.stream(self.streams[j]):
            self.qs[j].tt_moments(self.tmaps[i % len(self.tmaps)], self.mom_n[j], self.mom_S[j], self.mom_C[j])

    def run(self, first, count):
        """`count` reconstructions starting at step index `first`: in pair mode two steps (maps 0 and 1) per call."""
        if not self.pair:
            for i in range(first, first + count):
                self.step(i)
            return
        M = len(self.tmaps)
        for c in range(count // 2):
            k = first // 2 + c
            j = k % self.ns
            with self.torch.cuda.stream(self.streams[j]):
                self.qs[j].tt_moments2(self.tmaps[(2 * k) % M], self.tmaps[(2 * k + 1) % M], self.mom_n[j], self.mom_S[j], self.mom_C[j])
        if count & 1:
            self.step(first + count - 1)

    def bandpowers(self, which=0):
        """bandpowers of map `which` through this runner's path (public fine-grained calls, same kernels)"""
        q, P, e = self.q, self.P, self.eng
        kk = self.kks[0]
        q.reconstruct_tt_from_map(self.tmaps[which], out=kk)
        sums, _ = e.bin_power(kk, kk, self.norm, P["ids"], P["nids"], herm=True, active_cols=q.kappa_cols, active_rows=q.kappa_rows)
        self.torch.cuda.synchronize()
        return sums[1:-1] / self.counts[1:-1]

    def zero(self):
        for j in range(self.ns):
            self.mom_n[j].zero_(); self.mom_S[j].zero_(); self.mom_C[j].zero_()

    def rate(self, nsteps, nwarm=10):
        self.run(0, nwarm)
        self.torch.cuda.synchronize()
        t0 = time.perf_counter()
        self.run(0, nsteps)
        self.torch.cuda.synchronize()
        return nsteps / (time.perf_counter() - t0)


def side_count(args):
    """reconstructions per side measurement: the headline's K x B, bounded (the dense leg runs at 1.7 ms each)"""
    return int(min(max(args.steps * max(1, args.batch), 40), 400))


def side_leg(torch, args, name, ref_p1d, seed, **kw):
    """One side measurement: its own estimator + runner on the same synthetic job; bandpowers compared with `ref_p1d`
    (same maps: the GRF draw depends only on (seed, index))."""
    P = build_pipeline(args.n, args.res, kw.pop("prec", args.prec), torch, **kw)
    tm = make_maps(P, torch, seed)
    R = Runner(P, torch, tm, args.streams)
    rate = R.rate(side_count(args))
    p1d = R.bandpowers(0)
    out = {"reconstructions_per_s": rate, "streams_per_gpu": R.ns}
    if ref_p1d is not None:
        out["max_rel_bandpower_diff"] = float((p1d.double() / ref_p1d.double() - 1).abs().max().item())
    return out, p1d, P, R


def time_kernel(torch, fn, reps=20, warm=3):
    """Mean duration (s) of `fn`'s launches on the CURRENT stream (HIP events recorded on that stream)."""
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
    return e0.elapsed_time(e1) / reps * 1e-3


# --------------------------------------------------------------------------------------------------------------
# executed arithmetic of the fused row stage (tools/count_flops: the kernel body run on the host over a counting
# scalar type -- adds/multiplies with non-constant operands, packed-asm complex products as 6, a+-ib as 2)
# --------------------------------------------------------------------------------------------------------------
def row_qe_flops(N, win, wout, mrow=0):
    """(flops per launch, row grid, provenance).  Falls back to the nominal 5 L log2 L radix-2 count x 5 transforms."""
    W = N // 2 + 1
    win = W if not win else win
    wout = W if not wout else wout
    exe = os.path.join(ROOT, "tools", "_bin", "count_flops")
    if os.path.exists(exe):
        try:
            out = subprocess.run([exe, str(N), str(win), str(wout), str(mrow)], capture_output=True, text=True, timeout=300)
            d = json.loads(out.stdout.strip().splitlines()[-1])
            return float(d["flops_per_row"]) * N, int(d["mrow"]), "tools/count_flops (executed arithmetic of the kernel body, per row x rows)"
        except Exception:
            pass
    M = N
    if mrow < 0:
        M = 64
        while M < 2 * win + wout and M < N:
            M *= 2
        if 1024 < 2 * win + wout <= 1536 and win <= 512 and N >= 2048:
            M = 1536
    elif mrow > 0:
        M = mrow
    L = M // 2
    return N * 5 * 5.0 * L * np.log2(L), M, "nominal 5 L log2 L per packed transform x 5 (count_flops not built: upper bound)"


def load_profile_json(name):
    path = os.path.join(ROOT, "profiles", name)
    if os.path.exists(path):
        try:
            return json.load(open(path))
        except Exception:
            return None
    return None


def per_kernel_table(torch, P, R, args):
    """Live per-kernel timings (HIP events on the launch stream) with the bytes each kernel must move once
    (inputs + outputs on its active columns / rows) and, for the row stage, the arithmetic it executes."""
    q, eng = P["q"], P["eng"]
    N = eng.ny
    es = 4 if P["prec"] == "f32" else 8
    A = es * N * N
    W = N // 2 + 1
    Ah = 2 * es * N * W
    wl, wk = q.leg_cols, q.kappa_cols
    rl, rk = q.leg_rows, q.kappa_rows
    fl, fk = (wl or W) / float(W), (wk or W) / float(W)
    gl = (2 * rl - 1) / float(N) if rl else 1.0       # active row fractions
    gk = (2 * rk - 1) / float(N) if rk else 1.0
    from orphics_amd.engine import _ptr, _stream
    from orphics_amd._lib import check
    r1 = R.tmaps[0]
    e = q._bind_bins()                       # this handle's plan knows its filters and bins
    lib, plan = e.lib, e.plan

    def stage(k):
        # one stage of oa_qe_tt_moments on the plan's own (compact) work planes: the launches of the headline path
        return lambda: check(lib.oa_qe_tt_stage(plan, k, _ptr(r1), _stream()))
    scratch_moments = (R.mom_n[0].clone(), R.mom_S[0].clone(), R.mom_C[0].clone())     # fill every work plane once
    check(lib.oa_qe_tt_moments(plan, _ptr(r1), *[_ptr(t) for t in scratch_moments], _stream()))
    # column grid: the legs / row stage / divergence run on my of the N rows (include/orphics_amd.h, COLUMN GRID)
    my = int(lib.oa_plan_col_grid(plan))
    cg = my / float(N) if my else 1.0
    # name -> (stage, bytes it must move once: inputs + outputs on its active columns / rows)
    logn = int(round(np.log2(N)))
    rsplit = int(lib.oa_plan_rsplit(plan))
    if rsplit:
        legs_name = ("legs_cols = col_fband_kernel (single pass: forward %d-point columns of the R-split row output + leg filters + "
                     "inverse %d-point columns -> 3 leg planes on the %d-row column grid)" % (my, my // rsplit, my))
        # read the row pass's output once (R planes = ny rows) + 2 real filter planes on the band rows, write 3 planes of my rows
        legs_bytes = fl * (Ah + gl * Ah + 3 * cg * Ah)
    elif my and logn // 2 == 6 and (my >> (logn - 6)) == 16:
        legs_name = "fwdlegs_cols = col_fwdlegs_cg_kernel (fwd pass2 + legs + inv pass1) + col_fft_kernel<inv pass2 x3> on the %d-row column grid" % my
        # fused kernel: read the pass-1 plane + 2 real filter planes on the band rows, write 3 planes of my rows; then the
        # 3-plane inverse pass 2 on my rows (r + w)
        legs_bytes = fl * (Ah + gl * Ah + 3 * cg * Ah) + 6 * fl * cg * Ah
    elif my:
        legs_name = "fwdlegs_cols = col_fft_kernel<fwd pass2> + col_legs_kernel + col_fft_kernel<inv pass2 x3> on the %d-row column grid" % my
        # forward pass 2 (read the pass-1 plane, write the band rows) + col_legs (read the band rows + 2 real filter planes there,
        # write 3 planes of my rows) + 3-plane inverse pass 2 on my rows (r + w)
        legs_bytes = fl * (Ah + gl * Ah) + fl * (gl * Ah + gl * Ah + 3 * cg * Ah) + 6 * fl * cg * Ah
    else:
        legs_name = "fwdlegs_cols = col_fwdlegs_kernel + col_fft_kernel<inv pass2 x3>"
        # col_fwdlegs (read the pass-1 plane + 2 real filter planes on the band rows, write 3) + 3-plane inverse pass 2 (r + w)
        legs_bytes = fl * (Ah + gl * Ah + 3 * Ah) + 6 * fl * Ah
    if my in (1024, 2048):
        # single pass: read the 2 product planes of my rows + Fn/2 on the band rows, write the band rows of kappa_hat
        div_name, div_bytes = "cols_div = col_div_sp_kernel (single-pass forward columns + divergence)", fk * (2 * Ah * cg + gk * Ah / 2 + gk * Ah)
    else:
        # one 2-plane pass-1 launch (read 2, write 2) + col_div (read 2 + Fn/2, write the band rows of 1)
        div_name, div_bytes = "cols_div = col_fft_kernel<pass1 x2> + col_div_kernel", 4 * fk * Ah * cg + fk * (2 * Ah * cg + gk * Ah / 2 + gk * Ah)
    kern = {
        "row_fft_kernel<R2C>": (0, A + fl * Ah),
        "col_fft_kernel<fwd pass1, leg width>": (1, 2 * fl * Ah),
        legs_name: (2, legs_bytes),
        "row_qe_kernel": (3, (3 * fl + 2 * fk) * Ah * cg),
        div_name: (4, div_bytes),
        "bin_kernel<power>": (5, 1.5 * fk * gk * Ah),
    }
    if rsplit:           # the R-split row pass carries the first column radix: there is no separate forward column pass 1
        del kern["col_fft_kernel<fwd pass1, leg width>"]
    if int(lib.oa_plan_div_fused(plan)):
        # binning + moments ride in the tail of the divergence launch (csrc/fft_divbin.hpp): no histogram launches, kappa_hat is
        # neither written nor re-read: read the 2 product planes + Fn/2 + the int32 ids (= Ah/4 per full half plane) on the band rows
        del kern["bin_kernel<power>"]
        del kern[div_name]
        div_name = "cols_div_bin = col_div_sp_bin_kernel (single-pass forward columns + divergence + radial binning + moments)"
        kern[div_name] = (4, fk * (2 * Ah * cg + gk * Ah / 2 + gk * Ah / 4))
    # Stage durations IN SEQUENCE: whole steps (stages 0..5 back to back on this stream, alternating between the two
    # input maps as the timed loop does) with a HIP event between consecutive stages.  Timing one stage in a tight
    # loop of its own would let its inputs sit in the 256 MB infinity cache / L2 (the 268 MB map re-read 20 times
    # measured 49 us for the row pass against 60 us inside the step) -- these are the durations rocprofv3 reports for
    # the same kernels inside the timed loop (profiles/<tag>_step.txt).
    reps, warm = 24, 4
    names = sorted(kern, key=lambda n: kern[n][0])
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)] for _ in range(reps)]
    for it in range(warm + reps):
        rmap = _ptr(R.tmaps[it & 1])
        for si, name in enumerate(names):
            if it >= warm:
                evs[it - warm][si].record()
            check(lib.oa_qe_tt_stage(plan, kern[name][0], rmap, _stream()))
        if it >= warm:
            evs[it - warm][len(names)].record()
    torch.cuda.synchronize()
    per = {}
    for si, name in enumerate(names):
        dt = float(np.median([evs[r][si].elapsed_time(evs[r][si + 1]) for r in range(reps)])) * 1e-3
        moved = kern[name][1]
        per[name] = {"avg_ms": dt * 1e3, "hbm_min_GB": moved / 1e9, "hbm_GBs": moved / dt / 1e9, "hbm_frac": moved / dt / 1e9 / HBM_PEAK_GBS}
    flops, mrow, how = row_qe_flops(N, wl, wk, q.mrow)
    flops *= cg                                   # the row stage visits my of the N rows
    rq = per["row_qe_kernel"]
    rq.update({"executed_GFLOP": flops / 1e9, "flop_count": how, "row_grid": mrow, "TFLOPs": flops / (rq["avg_ms"] * 1e-3) / 1e12,
               "valu_frac": flops / (rq["avg_ms"] * 1e-3) / 1e12 / (VALU_PEAK_TFLOPS if P["prec"] == "f32" else VALU_PEAK_TFLOPS / 2.0),
               "valu_peak_TFLOPs": VALU_PEAK_TFLOPS if P["prec"] == "f32" else VALU_PEAK_TFLOPS / 2.0,
               "arithmetic_intensity_flop_per_B": flops / (rq["hbm_min_GB"] * 1e9)})
    return per, dict(A=A, Ah=Ah, fl=fl, fk=fk, W=W, wl=wl, wk=wk, rl=rl, rk=rk, mrow=mrow, mcol=my, rsplit=rsplit)


# --------------------------------------------------------------------------------------------------------------
# CPU baseline (scipy.fft workers = 1 and = nproc, one warm-up + median of 5, CPU model stated)
# --------------------------------------------------------------------------------------------------------------
def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


class _Sections:
    """wall time of every section of a bench run (reported as `sections_s`: what the driver's 10 minute limit is spent on)"""
    def __init__(self):
        self.t = {}
        self._t0 = time.perf_counter()

    def mark(self, name):
        now = time.perf_counter()
        self.t[name] = round(self.t.get(name, 0.0) + now - self._t0, 2)
        self._t0 = now


SECTIONS = _Sections()


def cpu_baseline(N_gpu, res_arcmin, reps=3):
    """NumPy/SciPy oracle (float64, full-plane C2C like the reference): on every host core AT THE WORKLOAD'S SIZE (no scaling: about
    5 s per 8192^2 reconstruction on the GPU box's 256 cores), and on one core on a bounded 2048^2 sample scaled by N^2 log N."""
    from oracle import maps_oracle as mo
    from oracle import qe_oracle as qo
    from oracle import stats_oracle as so
    cores = os.cpu_count() or 1
    res = res_arcmin * np.pi / 180. / 60.

    def one(Ns, workers, reps=reps):
        mo.set_workers(workers)
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
        ts = []
        for r in range(reps + 1):                       # first pass = warm-up (pocketfft plans, page faults)
            t0 = time.perf_counter()
            kT = fc.fft(tmap)
            kk = q.kappa_ft(kT)
            p2d = fc.f2power(kk, kk)
            binner.bin(p2d)
            ts.append(time.perf_counter() - t0)
        ts = ts[1:]
        scale = (N_gpu / Ns) ** 2 * (np.log2(float(N_gpu)) / np.log2(float(Ns)))
        return {"sample_side": Ns, "median_s": float(np.median(ts)), "min_s": float(np.min(ts)), "runs": reps,
                "scale_to_workload": scale, "reconstructions_per_s": 1.0 / (float(np.median(ts)) * scale)}
    n_all = N_gpu if (cores >= 32 and N_gpu <= 8192) else min(N_gpu, 4096 if cores >= 32 else 2048)
    n_one = min(N_gpu, 2048 if cores >= 32 else 1024)
    all_c = one(n_all, cores, 3)
    one_c = one(n_one, 1, 2)
    import scipy
    return {"value": all_c["reconstructions_per_s"], "unit": "reconstructions/s", "cores": cores, "kind": "port",
            "sample": "oracle (float64 full-plane C2C) TT reconstruction + binned auto-power: %dx%d on %d scipy.fft workers, "
                      "%dx%d on 1 worker; one warm-up then the median of 3 / 2 runs; scaled by N^2 log N to %dx%d where the sample is smaller (scale %.2f / %.2f)"
                      % (n_all, n_all, cores, n_one, n_one, N_gpu, N_gpu, all_c["scale_to_workload"], one_c["scale_to_workload"]),
            "workers_all": all_c, "workers_1": one_c, "value_workers_1": one_c["reconstructions_per_s"],
            "cpu_model": cpu_model(), "numpy": np.__version__, "scipy": scipy.__version__}


def lensed_loop_leg(torch, args, side=4096, nsims=12, estimators=("TT", "EB")):
    """SURVEY 3.4 / tutorials/tt_verification.ipynb cell 4, the north-star loop itself: FlatLensingSims.get_sim (3 GRFs +
    order-5 flat-sky lensing + beam + noise) -> T,E,B transforms -> TT and EB kappa_hat -> cross power with the input kappa
    -> 19 bandpowers -> device-side Statistics; simulations/s on one GPU with the per-stage split (HIP events)."""
    from orphics_amd import cosmology, lensing, maps, mc
    from orphics_amd.geometry import FlatGeometry
    prec = args.prec
    shape = (3, side, side)
    geom = FlatGeometry.from_res(shape, args.res)
    theory = cosmology.default_theory()
    sims = lensing.FlatLensingSims(shape, geom, theory, 1.5, 1.0, pol=True, dtype=prec)
    keep = {k: maps.mask_kspace(shape, geom, lmin=lo, lmax=hi) for k, (lo, hi) in (("T", (300., 2000.)), ("K", (20., 3500.)))}
    q = lensing.qest(shape, geom, theory, noise2d=sims.ps_noise[0, 0], beam2d=sims.kbeam, kmask=keep["T"], noise2d_P=sims.ps_noise[1, 1],
                     kmask_P=keep["T"], kmask_K=keep["K"], pol=True, unlensed_equals_lensed=True, dtype=prec)
    drv = mc.LensedSimsMonteCarlo(sims, q, np.linspace(20, 3500, 20), estimators=estimators)
    drv.run_local(range(2))                                # warm-up: estimator set-up (A_L), plans, code objects
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    drv.run_local(range(2, 2 + nsims))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / nsims
    drv.run_local(range(2 + nsims, 4 + nsims), stage_times=True)
    return {"sims_per_s": 1.0 / dt, "ms_per_sim": dt * 1e3, "side": side, "res_arcmin": args.res, "dtype": prec, "estimators": list(estimators),
            "nsims_timed": nsims, "stage_ms_per_sim": drv.stage_ms,
            "note": "mc.LensedSimsMonteCarlo: get_sim (unlensed T,Q,U + kappa + noise GRFs, order-5 flat-sky lensing of 3 maps, beam) -> "
                    "T,E,B -> TT + EB reconstructions -> cross / auto bandpowers -> device-side Statistics; one HIP stream"}


# --------------------------------------------------------------------------------------------------------------
# Legs of the other BASELINE configs at the reference's precision AND in float32: config 3 (8192^2 MV),
# config 4 (4096^2 Gaussian N0 + mean-field Monte Carlo, unwindowed and windowed), and the kappa-producing TT entry
# --------------------------------------------------------------------------------------------------------------
def _timeit(torch, fn, n, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def kappa_out_leg(torch, args, R):
    """The reference's contract ``kappa_from_map(..., returnFt=True)`` (lensing.py:973-976): oa_qe_tt from the real map with
    kappa_hat's transform WRITTEN to an estimator-owned plane (the headline's moment entries never store it), issued like the
    headline: resident maps round-robin over the runner's streams / handles."""
    M, ns = len(R.tmaps), R.ns
    count = int(min(max(args.steps * max(1, args.batch), 40), 640))

    def run():
        for i in range(count):
            j = i % ns
            with torch.cuda.stream(R.streams[j]):
                R.qs[j].reconstruct_tt_from_map(R.tmaps[i % M], out=R.kks[j])
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / count
    # the stored plane of the last call on stream 0, binned by the public histogram call, against the one-call bandpowers
    last0 = ((count - 1) // ns) * ns
    kk = R.kks[0]
    e, P, q = R.eng, R.P, R.q
    sums, _ = e.bin_power(kk, kk, R.norm, P["ids"], P["nids"], herm=True, active_cols=q.kappa_cols, active_rows=q.kappa_rows)
    got = (sums[1:-1] / R.counts[1:-1]).double()
    n, S, C = R.mom_n[0].clone().zero_(), R.mom_S[0].clone().zero_(), R.mom_C[0].clone().zero_()
    q.tt_moments(R.tmaps[last0 % M], n, S, C)
    torch.cuda.synchronize()
    return {"reconstructions_per_s": 1.0 / dt, "streams_per_gpu": ns, "reconstructions_timed": count,
            "max_rel_bandpower_diff_vs_moment_entry": float((got / S - 1).abs().max().item()),
            "note": "oa_qe_tt (Estimator.reconstruct_tt_from_map): real map -> kappa_hat DFT stored in an estimator-owned hc plane"}


def mv_leg(torch, args, N=8192, reps=10):
    """BASELINE config 3: 8192^2 five-estimator minimum-variance reconstruction (one oa_qe_mv call) in float64 and float32;
    bandpowers of the two precisions compared."""
    from orphics_amd import cosmology, lensing, maps
    from orphics_amd.geometry import FlatGeometry
    shape = (N, N)
    g = FlatGeometry.from_res(shape, args.res)
    th = cosmology.default_theory()
    nxh = N // 2
    ly, lx = g.laxes()
    ml_h = np.sqrt(ly[:, None] ** 2 + lx[None, :nxh + 1] ** 2)

    def full(a_h):
        o = np.empty(shape, dtype=a_h.dtype)
        o[:, :nxh + 1] = a_h
        o[:, nxh + 1:] = a_h[(-np.arange(N)) % N][:, 1:nxh][:, ::-1]
        return o
    beam_h = maps.gauss_beam(ml_h, 1.5)
    nT = cosmology.white_noise_power(1.0)
    noise = np.full(shape, nT)
    tmask = full(((ml_h > 300) & (ml_h < 2000)).astype(np.int64))
    kmask = full(((ml_h > 20) & (ml_h < 3500)).astype(np.int64))
    q64 = lensing.qest(shape, g, th, noise2d=noise, beam2d=full(beam_h), kmask=tmask, kmask_P=tmask, noise2d_P=2 * noise, kmask_K=kmask,
                       pol=True, unlensed_equals_lensed=True, dtype="f64")
    e64 = q64.eng
    ks64 = []
    for i, (sp, nz) in enumerate((("TT", nT), ("EE", 2 * nT), ("BB", 2 * nT))):
        cs = e64.hcreal()
        cs[:, :nxh + 1] = torch.as_tensor(np.sqrt((th.lCl(sp, ml_h) * beam_h ** 2 + nz) * float(N * N) ** 2 / g.area), device=e64.device)
        ks64.append(e64.grf_hc(77, i, cs))
        del cs
    edges = np.linspace(20, 3500, 20)
    out, bp = {}, {}
    for prec in ("f64", "f32"):
        q = q64 if prec == "f64" else q64.astype("f32")
        e = q.eng
        ks = [k.to(e.cdt) for k in ks64]
        own = q.new_output()
        q.reconstruct_mv_hc(*ks, out=own)
        dt = _timeit(torch, lambda: q.reconstruct_mv_hc(*ks, out=own), reps)
        ids = e.modl_digitize(torch.as_tensor(edges, device=e.device), half=True)
        s, c = e.bin_power(own, own, g.area / float(N * N) ** 2, ids, len(edges) + 1, herm=True)
        bp[prec] = (s[1:-1] / c[1:-1].double()).cpu().numpy()
        npieces = sum(len(q._gen[x]["pieces"]) for x in ("TT", "TE", "EE", "EB", "TB") if x in q._gen)
        out[prec] = {"mv_reconstructions_per_s": 1.0 / dt, "ms_per_mv_reconstruction": dt * 1e3, "separable_pieces": npieces}
        del ks, own
        if prec == "f32":
            del q
        torch.cuda.empty_cache()
    out["max_rel_bandpower_diff_f32_vs_f64"] = float(np.max(np.abs(bp["f32"] / bp["f64"] - 1)))
    out["config"] = "BASELINE config 3: %dx%d %.2f' TT+TE+EE+EB+TB minimum-variance combination, one oa_qe_mv call per reconstruction into an estimator-owned plane, one stream" % (N, N, args.res)
    return out


def mc_leg(torch, args, N=4096, nsims=480):
    """BASELINE config 4 per GPU: Gaussian N0 (+ mean-field stack) Monte Carlo at 4096^2 through oa_mc_run, unwindowed and with
    the reference's apodisation taper (oa_mc_run_windowed), float64 and float32; Monte-Carlo N0 against the analytic N_L."""
    from orphics_amd import cosmology, lensing, maps, mc, stats
    from orphics_amd.geometry import FlatGeometry
    shape = (N, N)
    g = FlatGeometry.from_res(shape, args.res)
    th = cosmology.default_theory()
    ml = g.modlmap()
    beam = maps.gauss_beam(ml, 1.5)
    noise = np.full(shape, cosmology.white_noise_power(1.0))
    tmask = ((ml > 300) & (ml < 2000)).astype(np.int64)
    kmask = ((ml > 20) & (ml < 3500)).astype(np.int64)
    q64 = lensing.qest(shape, g, th, noise2d=noise, beam2d=beam, kmask=tmask, kmask_K=kmask, unlensed_equals_lensed=True, dtype="f64")
    tot = (th.lCl("TT", ml) * beam ** 2 + noise)[:, :N // 2 + 1]
    edges = np.linspace(20, 3500, 20)
    _, nl = stats.bin2D(ml, edges).bin(q64.N_kappa("TT"))
    taper, w2 = maps.get_taper(shape, g)
    out, means = {}, {}
    for prec in ("f64", "f32"):
        q = q64 if prec == "f64" else q64.astype("f32")
        blk = {}
        for key, kw, ns_ in (("n0", dict(mean_field=False), nsims), ("n0_mean_field", dict(mean_field=True), nsims),
                             ("windowed_n0_mean_field", dict(mean_field=True, window=taper), max(60, nsims // 4))):
            drv = mc.GaussianN0MonteCarlo(q, tot, edges, comm=None, base_seed=1234, **kw)
            drv.run_local(range(12))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            drv.run_local(range(12, 12 + ns_))
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / ns_
            blk[key] = {"sims_per_s": 1.0 / dt, "ms_per_sim": dt * 1e3, "sims_timed": ns_}
            if key == "n0":
                drv.acc.allreduce()                     # (single rank: makes the reduced view)
                m = drv.acc.mean("n0")
                sem = np.sqrt(drv.acc.var("n0") / drv.acc.count("n0"))
                means[prec] = m
                blk[key]["max_abs_pull_vs_analytic_N0"] = float(np.max(np.abs((m - nl) / sem)))
                blk[key]["max_rel_dev_vs_analytic_N0"] = float(np.max(np.abs(m / nl - 1)))
            del drv
            torch.cuda.empty_cache()
        out[prec] = blk
        if prec == "f32":
            del q
    out["max_rel_diff_mean_bandpowers_f32_vs_f64"] = float(np.max(np.abs(means["f32"] / means["f64"] - 1)))
    out["window"] = {"kind": "maps.get_taper default cosine taper", "mean_w2": float(w2)}
    out["config"] = ("BASELINE config 4, one GPU's shard: %dx%d %.2f' Gaussian realisations (Philox, key = (seed, index)) -> TT estimator -> 19 "
                     "bandpowers -> device-side moments [+ mean-field stack], oa_mc_run / oa_mc_run_windowed, one stream" % (N, N, args.res))
    return out



def run_mc_config(args, torch, dist, world, rank):
    """``bench.py --gpus N --config mc``: BASELINE config 4 as the sharded job it is -- ``--mc-sims`` Gaussian realisations at
    ``--mc-n``^2 split over the ranks by the reference's rule (mpi.mpi_distribute: contiguous blocks, remainder on the last
    ranks), every rank's shard ONE oa_mc_run call with device-resident moments + mean-field stack, then Statistics.allreduce
    (one packed all-reduce of the moments + the region-only reduce of the mean-field stack).  Reports the whole-job rate
    (barrier + synchronize on both sides, MAX over ranks), every rank's compute time and the reduce time separately."""
    from orphics_amd import cosmology, lensing, maps, mc, mpi, stats
    from orphics_amd.geometry import FlatGeometry
    N, nsims, prec = args.mc_n, args.mc_sims, args.prec
    shape = (N, N)
    g = FlatGeometry.from_res(shape, args.res)
    th = cosmology.default_theory()
    ml = g.modlmap()
    beam = maps.gauss_beam(ml, 1.5)
    noise = np.full(shape, cosmology.white_noise_power(1.0))
    q = lensing.qest(shape, g, th, noise2d=noise, beam2d=beam, kmask=((ml > 300) & (ml < 2000)).astype(np.int64),
                     kmask_K=((ml > 20) & (ml < 3500)).astype(np.int64), unlensed_equals_lensed=True, dtype=prec)
    tot = (th.lCl("TT", ml) * beam ** 2 + noise)[:, :N // 2 + 1]
    edges = np.linspace(20, 3500, 20)
    comm = mpi.TorchComm() if world > 1 else None
    window = maps.get_taper(shape, g)[0] if args.mc_windowed else None
    # warm-up on a throw-away driver: code objects, plans, batch planes, and one rehearsal of the reduction (first use of the
    # communicator opens its connections)
    warm = mc.GaussianN0MonteCarlo(q, tot, edges, comm=comm, base_seed=99, mean_field=True, window=window)
    warm.run_local(range(rank * 16, rank * 16 + 16))
    warm.acc.allreduce()
    torch.cuda.synchronize()
    del warm
    drv = mc.GaussianN0MonteCarlo(q, tot, edges, comm=comm, base_seed=1234, mean_field=True, window=window)
    _, tasks = mpi.mpi_distribute(nsims, max(world, 1), allow_empty=True)
    mine = tasks[rank]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    drv.run_local(mine)
    torch.cuda.synchronize()
    t_compute = time.perf_counter() - t0
    drv.acc.allreduce()
    torch.cuda.synchronize()
    t_reduce = time.perf_counter() - t0 - t_compute
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [(rank, len(mine), t_compute, t_reduce)]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=q.eng.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered
    st = drv.acc
    assert st.count("n0") == nsims and st.stack_count("mf") == nsims, (st.count("n0"), st.stack_count("mf"), nsims)
    if rank != 0:
        return None
    _, nl = stats.bin2D(ml, edges).bin(q.N_kappa("TT"))
    mean = drv.debiased_mean()
    sem = np.sqrt(st.var("n0") / nsims) / drv.window_moments[1]
    return {"metric": "Monte-Carlo N0 + mean-field simulations/sec on %d^2 maps" % N, "value": nsims / elapsed, "unit": "simulations/s",
            "n_gpus": world, "steps": 1, "warmup": 1, "ms_per_step": elapsed * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": prec, "data": "synthetic",
            "config": {"workload": "BASELINE config 4: %d-simulation Gaussian N0 + mean-field Monte Carlo on %dx%d %.2f-arcmin maps%s, "
                                   "simulations sharded by mpi_distribute, one ensemble reduce" % (nsims, N, N, args.res, ", 12 % cosine taper" if window is not None else ""),
                       "map_side": N, "nsims": nsims, "windowed": window is not None, "parallelism": "sims sharded over %d rank(s), no data-path collective; "
                       "Statistics.allreduce at the end (packed moments + mean-field stack on kappa's active region)" % max(world, 1)},
            "per_rank": [{"rank": r, "sims": n_, "compute_s": tc, "reduce_s": tr} for (r, n_, tc, tr) in per_rank],
            "reduce_s_rank0": t_reduce, "elapsed_s": elapsed,
            "check": {"max_rel_dev_vs_analytic_N0": float(np.max(np.abs(mean / nl - 1))), "max_abs_pull": float(np.max(np.abs((mean - nl) / sem))),
                      "sims_counted": int(st.count("n0")), "stacked": int(st.stack_count("mf"))}}


def bandwidth_ceiling(torch, gb=2.0, reps=12):
    """What a plain streaming kernel reaches on THIS box in THIS run (16-byte accesses, HIP events on the launch stream):
    device copy (read + write) and read-only, GB/s over `gb` GB buffers (larger than the 256 MB infinity cache)."""
    from orphics_amd import _lib
    from orphics_amd.engine import _ptr, _stream
    lib = _lib.load()
    n = int(gb * 1e9) // 16 * 16
    a = torch.empty(n, dtype=torch.uint8, device="cuda")
    b = torch.empty(n, dtype=torch.uint8, device="cuda")
    a.zero_(); b.zero_()
    sink = torch.empty(8 << 20, dtype=torch.uint8, device="cuda")
    t_copy = time_kernel(torch, lambda: _lib.check(lib.oa_probe_copy(_ptr(b), _ptr(a), n, _stream())), reps=reps)
    t_read = time_kernel(torch, lambda: _lib.check(lib.oa_probe_read(_ptr(a), n, _ptr(sink), _stream())), reps=reps)
    del a, b, sink
    torch.cuda.empty_cache()
    return {"copy_GBs": 2.0 * n / t_copy / 1e9, "read_GBs": n / t_read / 1e9, "buffer_GB": n / 1e9,
            "how": "oa_probe_copy / oa_probe_read: 16-byte grid-stride accesses, mean of %d launches timed with HIP events in this run" % reps}


# --------------------------------------------------------------------------------------------------------------
def map_usage(nsteps, nmaps, pair):
    """how often each resident map is reconstructed by Runner.run(0, nsteps) (the timed region)"""
    use = np.zeros(nmaps, dtype=np.int64)
    if not pair:
        for i in range(nsteps):
            use[i % nmaps] += 1
        return use
    for c in range(nsteps // 2):
        use[(2 * c) % nmaps] += 1
        use[(2 * c + 1) % nmaps] += 1
    if nsteps & 1:
        use[(nsteps - 1) % nmaps] += 1
    return use


def measure(args, torch, dist, world, rank, prec):
    """The job in one precision: resident batch, pre-roll, warm-up, EXACTLY K timed steps bracketed by barrier +
    synchronize, moment-counter and bandpower checks, then (rank 0) the per-kernel table and the roofline object."""
    N = args.n
    P = build_pipeline(N, args.res, prec, torch, prune=not args.no_prune, tlmax=args.tlmax, row_grid=args.row_grid)
    q, eng = P["q"], P["eng"]
    es = 4 if prec == "f32" else 8
    seed = 1234 + rank                                    # distinct realisations per rank
    B = max(1, args.batch)
    # the batch is resident in HBM before the timed region: B distinct maps (fewer, cycled, if B of them would not leave
    # room: 16384^2 f64 maps are 2.1 GB each)
    nmaps = max(2, min(B, int(32e9 // (float(es) * N * N))))
    tmaps = make_maps(P, torch, seed, nmaps)
    if getattr(args, "auto_streams", False):
        # auto: three streams below 8192^2 and for the float64 kernels at 8192^2 and above, two for float32 there (A/B on one box,
        # profiles/r05_streams.txt: f64 5.25 / 5.39 / 5.22 k recon/s on 2 / 3 / 4 streams, f32 10.6 / 10.2 / 9.8 k)
        args.streams = 3 if (args.n < 8192 or prec == "f64") else 2
    R = Runner(P, torch, tmaps, args.streams, pair=not args.no_pair)
    ns = R.ns

    # pre-roll: a fresh box idles at ~550 MHz sclk and needs a few hundred ms of load to reach its sustained
    # clocks; W warm-up steps alone (~10 ms) would leave the ramp inside the timed region.  Untimed, uncounted.
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.preroll:
        R.run(0, 8)
        torch.cuda.synchronize()
    R.run(0, args.warmup * B)
    torch.cuda.synchronize()
    # rehearse the end-of-job reduction once (first use of a torch op / of the RCCL communicator loads code
    # objects and opens connections: tens of ms that belong to start-up, not to the K timed steps)
    wn, wS, wC = sum(R.mom_n), sum(R.mom_S), sum(R.mom_C)
    if world > 1:
        dist.all_reduce(wn); dist.all_reduce(wS); dist.all_reduce(wC)
    torch.cuda.synchronize()
    del wn, wS, wC
    R.zero()                                 # the timed region counts only its own realisations
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs = []
    if args.trace_steps:
        for i in range(args.steps * B):
            R.step(i)
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(R.streams[i % ns])
            evs.append(ev)
    else:
        R.run(0, args.steps * B)      # exactly K batches of B reconstructions (pair mode: two maps per C-ABI call)
    t_issue = time.perf_counter() - t0          # host time to enqueue the K steps (diagnostic: must stay < elapsed)
    torch.cuda.synchronize()
    if args.trace_steps and rank == 0:
        ts = [evs[0].elapsed_time(e) for e in evs]
        sys.stderr.write("event span first->last step end: %.2f ms; wall to issue %.2f ms\n" % (ts[-1], t_issue * 1e3))
        for a in range(0, len(ts) - 20, 20):
            sys.stderr.write("steps %4d-%4d: %.1f recon/s\n" % (a, a + 20, 20.0 / max(ts[a + 20] - ts[a], 1e-9) * 1e3))
    loc_n, loc_S = int(sum(R.mom_n).item()), sum(R.mom_S).clone()      # this rank's part, for the bandpower check below
    mom_n, mom_S, mom_C = sum(R.mom_n), sum(R.mom_S), sum(R.mom_C)     # per-stream accumulators
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
    assert total == args.steps * B * max(world, 1), "moment counter %d != steps x batch x ranks" % total
    # the timed region really produced the bandpowers: its accumulated sum equals the usage-weighted sum of the per-map
    # bandpowers computed through the fine-grained public calls (same kernels, one map at a time)
    use = map_usage(args.steps * B, len(tmaps), R.pair and not args.trace_steps)
    assert int(use.sum()) == loc_n
    check_maps = [i for i in range(len(tmaps)) if use[i]]
    if len(check_maps) > args.check_maps:        # bounded: a spread of the batch (the S comparison then uses the accumulator of a re-run)
        check_maps = check_maps[:: max(1, len(check_maps) // args.check_maps)][:args.check_maps]
    per_map = {i: R.bandpowers(i).double() for i in check_maps}
    if len(check_maps) == int((use > 0).sum()):
        want = sum(per_map[i] * float(use[i]) for i in check_maps)
        dev = float(((loc_S / want) - 1).abs().max().item())
        assert dev < (1e-9 if prec == "f64" else 2e-6), "timed-region bandpower sum differs from the per-map bandpowers: %g" % dev
    else:
        # one extra (untimed) pass over the checked maps through the SAME one-call entry
        R.zero()
        for i in check_maps:
            with torch.cuda.stream(R.streams[0]):
                R.qs[0].tt_moments(tmaps[i], R.mom_n[0], R.mom_S[0], R.mom_C[0])
        torch.cuda.synchronize()
        want = sum(per_map[i] for i in check_maps)
        dev = float(((R.mom_S[0] / want) - 1).abs().max().item())
        assert dev < (1e-9 if prec == "f64" else 2e-6), "one-call bandpowers differ from the fine-grained path: %g" % dev
    res = {"prec": prec, "P": P, "R": R, "tmaps": tmaps, "seed": seed, "total": total, "elapsed": elapsed, "t_issue": t_issue,
           "value": total / elapsed, "ms_per_step": elapsed / args.steps * 1e3, "bandpower_check": {"maps_checked": len(check_maps), "max_rel_dev": dev,
           "how": "accumulated S of the timed one-call path vs the usage-weighted per-map bandpowers of the fine-grained public calls"},
           "mean_bandpowers": (mom_S / float(total)).cpu().numpy().tolist()}
    if rank != 0:
        return res
    per, G = per_kernel_table(torch, P, R, args)
    A, W = G["A"], G["W"]
    share = {k: v["avg_ms"] for k, v in per.items()}
    dom = max(share, key=share.get)
    rate = total / elapsed / max(world, 1)
    traffic_tab = None
    if N == 8192 and args.tlmax == 2000.0:      # PMC traffic exists for the profiled configurations
        suf = "_dense" if args.no_prune else ("_fullrows" if args.row_grid == "full" else "")
        traffic_tab = load_profile_json("traffic_%s_%s%s.json" % (PROFILE_TAG, prec, suf))
    short = dom.split(" ")[0]
    traffic = (traffic_tab or {}).get(short)
    d = per[dom]
    vpeak = VALU_PEAK_TFLOPS if prec == "f32" else VALU_PEAK_TFLOPS / 2.0       # f64 vector rate = half the packed-f32 rate
    ridge = vpeak * 1e12 / (HBM_PEAK_GBS * 1e9)
    if "TFLOPs" in d and d["arithmetic_intensity_flop_per_B"] > ridge:
        roofline = {"bound": "valu", "kernel": dom, "achieved": d["TFLOPs"], "peak": vpeak, "unit": "TFLOP/s",
                    "frac": d["TFLOPs"] / vpeak, "traffic": traffic, "flops_per_launch": d["executed_GFLOP"] * 1e9,
                    "flop_count": d["flop_count"], "arithmetic_intensity_flop_per_B": d["arithmetic_intensity_flop_per_B"],
                    "ridge_flop_per_B": ridge, "hbm_bytes_per_launch": d["hbm_min_GB"] * 1e9, "hbm_frac_on_those_bytes": d["hbm_frac"],
                    "note": "the fused row stage (3 C2R + 2 products + 2 R2C per row in LDS/registers) moves %.2f GB per launch and executes "
                            "%.1f GFLOP: arithmetic intensity above the ridge -> priced against the %s vector peak on the arithmetic "
                            "it executes (pruned taps not counted)" % (d["hbm_min_GB"], d["executed_GFLOP"], prec)}
    else:
        roofline = {"bound": "hbm", "kernel": dom, "achieved": d["hbm_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": d["hbm_frac"], "traffic": traffic, "bytes_per_launch": d["hbm_min_GB"] * 1e9,
                    "note": "achieved = ALGORITHMIC bytes of the launch (inputs + outputs on its active columns/rows: the whole map read once + "
                            "the kept columns written) / live HIP-event duration inside the step sequence; traffic = PMC bytes of the same "
                            "launch from profiles/traffic_%s_%s.json (separate rocprofv3 --pmc runs of this command, not measured in this run)" % (PROFILE_TAG, prec)}
    assert roofline["frac"] <= 1.0, "roofline fraction %g > 1: byte/flop model is wrong" % roofline["frac"]
    if dom.startswith("row_fft_kernel<R2C>"):
        # the stage name above is bench.py's; the launch behind it, as rocprofv3 lists it (fft.hip HipLauncher::row_w64)
        wl_ = G["wl"] or W
        w64 = prec == "f32"
        tn = "float" if prec == "f32" else "double"
        if G.get("rsplit"):
            rs = (N == 16384 and wl_ <= 512 and prec == "f64") or (N == 8192 and wl_ <= 1280) or (N == 4096 and wl_ <= 256)
            wide = "w" if (N == 8192 and wl_ > 512) else ""       # the wide band (R = 2): row_r2c_rs4096w_kernel
            roofline["kernel_symbol"] = ("row_r2c_rs%d%s_kernel<%s, ...>" % (N // 2, wide, tn) if rs else "row_r2c_rsplit_kernel<%s, ...>" % tn)
            roofline["rsplit"] = {"R": G["rsplit"], "note": "the row pass also takes the first radix-R butterfly of the column transform (rows g + my n, n < R, "
                                  "per workgroup) and writes R planes Y[k1][g]; one single-pass column kernel follows (include/orphics_amd.h oa_plan_rsplit)"}
        else:
            roofline["kernel_symbol"] = ("row_r2c_w64_kernel" if (w64 and N == 8192 and wl_ <= 512) else
                                         "row_r2c_w64x2_kernel" if (w64 and N == 16384 and wl_ <= 768) else "row_fft_kernel<%s, R2C>" % tn)
    roofline["active_columns"] = {"legs": G["wl"] or W, "kappa": G["wk"] or W, "of": W}
    roofline["row_grid"] = {"points": G["mrow"], "of": N, "note": "band-limited legs: the real-space products are formed on the smallest "
                            "alias-free row grid >= 2 leg_cols + kappa_cols the row stage is built for -- 1024, 1536 (= 3 x 512), 2048, 4096, 8192 "
                            "points (exact; include/orphics_amd.h ROW GRID)"}
    roofline["col_grid"] = {"rows": G["mcol"] or N, "of": N, "note": "the same argument along y: inverse column transforms of the legs, row stage "
                            "and forward column transforms of the products run on the smallest alias-free power-of-two number of rows "
                            ">= max(2 leg_rows + kappa_rows, 2 kappa_rows) (exact; include/orphics_amd.h COLUMN GRID); extra.fullres_rows "
                            "is the same job with both grids at the map's own resolution"}
    roofline["active_rows"] = {"legs": (2 * G["rl"] - 1) if G["rl"] else N, "kappa": (2 * G["rk"] - 1) if G["rk"] else N, "of": N}
    roofline["share_of_recon_ms"] = share
    roofline["per_kernel"] = per
    roofline["pmc_traffic_bytes_per_launch"] = {k: v for k, v in (traffic_tab or {}).items() if not k.startswith("_")} or None
    if traffic_tab:
        for k, v in per.items():           # real (PMC) bytes / live duration for every kernel of the step
            b = traffic_tab.get(k.split(" ")[0])
            if b:
                v["pmc_GB"] = b / 1e9
                v["pmc_GBs"] = b / (v["avg_ms"] * 1e-3) / 1e9
                v["pmc_hbm_frac"] = v["pmc_GBs"] / HBM_PEAK_GBS
    hbm = {"peak_GBs": HBM_PEAK_GBS,
           "kernels_on_own_bytes": {k: {"GBs": v["hbm_GBs"], "frac": v["hbm_frac"]} for k, v in per.items() if "hbm_GBs" in v and k != "row_qe_kernel"},
           "survey_8d_bytes_per_recon_dense": 37.25 * A}
    if traffic_tab and traffic_tab.get("bytes_per_recon"):
        # every launch of a reconstruction together: PMC bytes through the L2 <-> fabric interface x reconstructions/s.
        # (Kernel boundaries flush the per-XCD L2s, so every intermediate plane makes the round trip even when the
        # infinity cache holds it.)
        bpr = float(traffic_tab["bytes_per_recon"])
        hbm["whole_pipeline_on_pmc_bytes"] = {"bytes_per_recon": bpr, "GBs": bpr * rate / 1e9, "frac": bpr * rate / 1e9 / HBM_PEAK_GBS,
                                              "pmc_source": "profiles/traffic_%s_%s.json (not measured in this run)" % (PROFILE_TAG, prec)}
    res.update({"roofline": roofline, "hbm": hbm, "G": G, "per": per, "A": A, "W": W})
    if world == 1 and not args.no_extras:
        try:
            res["kappa_out"] = kappa_out_leg(torch, args, R)
        except Exception as ex:          # a side leg never takes the headline down
            res["kappa_out"] = {"error": repr(ex)}
    return res


def block_of(args, res, world, dist):
    """the JSON fields of one precision's measurement"""
    R, P = res["R"], res["P"]
    N = args.n
    return {"value": res["value"], "unit": "reconstructions/s", "dtype": res["prec"], "ms_per_step": res["ms_per_step"],
            "host_issue_ms_per_step": res["t_issue"] / args.steps * 1e3, "reconstructions_timed": res["total"],
            "bandpower_check": res["bandpower_check"],
            "config": {"workload": "TT quadratic estimator (lensing.Estimator) on %dx%d %.2f-arcmin flat-sky GRF maps, "
                                   "incl. R2C of the input map and 19-bin kappa auto-bandpowers; T filter ell in (300,%d), "
                                   "kappa mask (20,3500), 1.5' beam, 1 uK' noise" % (N, N, args.res, int(args.tlmax)),
                       "map_side": N, "res_arcmin": args.res, "estimator": "TT", "nbins": R.d,
                       "maps_per_step": max(1, args.batch), "distinct_resident_maps": len(res["tmaps"]),
                       "streams_per_gpu": R.ns, "realisations_per_call": 2 if R.pair else 1,
                       "parallelism": "independent realisations per GPU + 1 all-reduce of bandpower moments"},
            "roofline": res.get("roofline"), "hbm": res.get("hbm"), "kappa_out": res.get("kappa_out")}


def release(res, torch):
    for k in ("R", "P", "tmaps"):
        res.pop(k, None)
    import gc
    gc.collect()
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40, help="timed steps; ONE STEP = one batch of --batch independent maps through the whole path")
    ap.add_argument("--warmup", type=int, default=4, help="untimed steps (batches) before the timed region")
    ap.add_argument("--batch", type=int, default=64, help="maps per step: a shard of independent realisations resident in HBM, each one "
                    "map -> kappa_hat -> bandpowers -> moments (SURVEY 8e: the unit a GPU is handed in the Monte-Carlo job)")
    ap.add_argument("--n", type=int, default=8192, help="map side (default 8192, the metric's size)")
    ap.add_argument("--res", type=float, default=0.5)
    ap.add_argument("--prec", default="f64", choices=["f32", "f64"], help="precision of the HEADLINE (value, roofline): f64 = the reference's "
                    "arithmetic (float64 maps, complex128 transforms: maps.py:1613)")
    ap.add_argument("--also", default="auto", choices=["auto", "none", "f32", "f64"], help="second, equally complete measurement reported as a "
                    "top-level block named after its precision (auto: the other precision)")
    ap.add_argument("--tlmax", type=float, default=2000.0, help="upper ell of the T filter (SURVEY 8d: 2000; high-res variant 6000)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-prune", action="store_true",
                    help="process all nx/2+1 columns of every plane even where the band-limited filters vanish")
    ap.add_argument("--trace-steps", action="store_true", help="stderr: throughput per 20 timed steps (diagnostic)")
    ap.add_argument("--preroll", type=float, default=1.5, help="seconds of untimed load before the warm-up steps (clock ramp)")
    ap.add_argument("--check-maps", type=int, default=64, help="resident maps whose bandpowers are recomputed through the fine-grained calls "
                    "and compared with the timed region's accumulated sum")
    ap.add_argument("--no-extras", action="store_true", help="skip the side legs reported under 'extra' (never the headline value)")
    ap.add_argument("--extras", default="fullres_rows,dense,wideband,lensed_loop,mv,mc", help="comma list of side legs to run")
    ap.add_argument("--row-grid", default="auto", choices=["auto", "full"],
                    help="grid of the fused row stage's real-space products: auto = smallest alias-free power of two "
                         "(exact for band-limited filters; library default), full = the map's nx points")
    ap.add_argument("--no-pair", action="store_true", help="one realisation per C-ABI call (oa_qe_tt_moments) instead of two (oa_qe_tt_moments2)")
    ap.add_argument("--streams", type=int, default=0, help="HIP streams: independent realisations are issued round-robin "
                    "on this many streams (each with its own plan/workspace) so latency-bound and bandwidth-bound kernels overlap; "
                    "0 = auto: 2 for sides >= 8192 (5205 vs 5114 recon/s with 3, f64), 3 below (4096^2 f32: 37.2 k vs 30.7 k with 2)")
    ap.add_argument("--config", default="qe", choices=["qe", "mc"], help="qe (default): the headline metric; mc: BASELINE config 4 as a "
                    "sharded job (--mc-sims realisations at --mc-n^2 split over --gpus ranks, one ensemble reduce), its own JSON line")
    ap.add_argument("--mc-n", type=int, default=4096)
    ap.add_argument("--mc-sims", type=int, default=1000)
    ap.add_argument("--mc-windowed", action="store_true", help="--config mc with the reference's apodisation taper (oa_mc_run_windowed)")
    args = ap.parse_args()
    args.auto_streams = args.streams <= 0
    if args.streams <= 0:
        args.streams = 2 if args.n >= 8192 else 3

    world, rank, local_rank, spawn = resolve_world(args.gpus, os.environ)
    if spawn:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

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
        assert dist.get_world_size() == args.gpus, "process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus)

    if args.config == "mc":
        line = run_mc_config(args, torch, dist, world, rank)
        if rank == 0:
            print(json.dumps(line))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    N = args.n
    other = {"auto": "f32" if args.prec == "f64" else "f64", "none": None}.get(args.also, args.also)
    if other == args.prec:
        other = None
    SECTIONS.mark("startup")
    head = measure(args, torch, dist, world, rank, args.prec)
    SECTIONS.mark("headline_" + args.prec)
    out = None
    if rank == 0:
        blk = block_of(args, head, world, dist)
        out = {"metric": "QE kappa reconstructions/sec on %d^2 maps" % N,
               "value": blk["value"], "unit": blk["unit"], "n_gpus": world, "world_size": (dist.get_world_size() if world > 1 else 1),
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": blk["ms_per_step"],
               "host_issue_ms_per_step": blk["host_issue_ms_per_step"], "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": args.prec, "data": "synthetic",
               "config": blk["config"], "roofline": blk["roofline"], "bandpower_check": blk["bandpower_check"]}
        hbm = blk["hbm"]
        if world == 1:
            hbm["measured_streaming_ceiling"] = bandwidth_ceiling(torch)
        A, W = head["A"], head["W"]
        R, P, tmaps, seed = head["R"], head["P"], head["tmaps"], head["seed"]
        if world == 1 and not args.no_extras and not args.no_prune:
            want = [w for w in args.extras.split(",") if w]
            ref_p1d = R.bandpowers(0)
            extra = {}
            G = dict(head["G"])
            if "fullres_rows" in want and G["mrow"] < N:
                leg, _, Pf, Rf = side_leg(torch, args, "fullres_rows", ref_p1d, seed, tlmax=args.tlmax, row_grid="full")
                perf, _ = per_kernel_table(torch, Pf, Rf, args)
                rqf = perf["row_qe_kernel"]
                leg["row_qe_kernel"] = {k: rqf[k] for k in ("avg_ms", "executed_GFLOP", "TFLOPs", "valu_frac", "row_grid")}
                leg["share_of_recon_ms"] = {k: v["avg_ms"] for k, v in perf.items()}
                leg["note"] = "row_grid='full': the fused row stage on all nx points of every row (the round-1 configuration)"
                extra["fullres_rows"] = leg
                del Pf, Rf, perf
                torch.cuda.empty_cache()
                SECTIONS.mark("fullres_rows")
            if "dense" in want:
                leg, _, _, _ = side_leg(torch, args, "dense", ref_p1d, seed, prune=False, tlmax=args.tlmax)
                rd = leg["reconstructions_per_s"]
                leg["note"] = "prune=False: all nx/2+1 columns of every plane are transformed (filters without a band limit)"
                leg["pipeline_GBs_on_survey_37.25A"] = 37.25 * A * rd / 1e9
                leg["pipeline_frac_of_hbm_peak_on_survey_37.25A"] = 37.25 * A * rd / 1e9 / HBM_PEAK_GBS
                dense_pmc = (load_profile_json("traffic_%s_%s_dense.json" % (PROFILE_TAG, args.prec)) or {}).get("bytes_per_recon") if N == 8192 else None
                if dense_pmc:
                    leg["pmc_bytes_per_recon"] = dense_pmc
                    leg["pipeline_GBs_on_pmc_bytes"] = dense_pmc * rd / 1e9
                    leg["pipeline_frac_of_hbm_peak_on_pmc_bytes"] = dense_pmc * rd / 1e9 / HBM_PEAK_GBS
                extra["dense"] = leg
                hbm["dense_pipeline"] = {k: leg[k] for k in leg if k.startswith("pipeline_") or k == "pmc_bytes_per_recon"}
                torch.cuda.empty_cache()
                SECTIONS.mark("dense")
            if "wideband" in want and args.tlmax < 6000.0:
                leg, p_w, Pw, Rw = side_leg(torch, args, "wideband", None, seed, tlmax=6000.0)
                leg["active_columns"] = {"legs": Pw["q"].leg_cols or W, "kappa": Pw["q"].kappa_cols or W, "of": W}
                leg["note"] = "SURVEY 8d high-res variant: T filter ell in (300,6000) on the same 0.5' maps"
                perw, _ = per_kernel_table(torch, Pw, Rw, args)
                leg["share_of_recon_ms"] = {k: v["avg_ms"] for k, v in perw.items()}
                leg["row_qe_valu_frac"] = perw["row_qe_kernel"]["valu_frac"]
                del Rw, Pw, perw
                torch.cuda.empty_cache()
                if other:
                    leg2, _, _, _ = side_leg(torch, args, "wideband_" + other, p_w, seed, prec=other, tlmax=6000.0)
                    leg["max_rel_bandpower_diff_vs_%s" % other] = leg2["max_rel_bandpower_diff"]
                    leg["%s_reconstructions_per_s" % other] = leg2["reconstructions_per_s"]
                extra["wideband"] = leg
                torch.cuda.empty_cache()
                SECTIONS.mark("wideband")
            if "lensed_loop" in want:
                try:
                    extra["lensed_loop"] = lensed_loop_leg(torch, args)
                except Exception as ex:      # a side leg never takes the headline down
                    extra["lensed_loop"] = {"error": repr(ex)}
                torch.cuda.empty_cache()
                SECTIONS.mark("lensed_loop")
            extra["kappa_out"] = blk.get("kappa_out")
            out["extra"] = extra
        out["hbm"] = hbm
    ref_head = head["R"].bandpowers(0).double().cpu() if rank == 0 else None
    release(head, torch)
    if other:
        sec = measure(args, torch, dist, world, rank, other)
        if rank == 0:
            blk2 = block_of(args, sec, world, dist)
            blk2["max_rel_bandpower_diff_vs_%s" % args.prec] = float((sec["R"].bandpowers(0).double().cpu() / ref_head - 1).abs().max().item())
            blk2["note"] = "the same job, same maps (the GRF draw depends only on (seed, index)), through the %s kernels: its own timed " \
                           "region of K steps, roofline and per-kernel table" % other
            out[other] = blk2
        release(sec, torch)
        SECTIONS.mark("second_block_" + other)
    if rank == 0:
        if world == 1 and not args.no_extras and not args.no_prune:
            want = [w for w in args.extras.split(",") if w]
            for name, fn in (("mv", mv_leg), ("mc", mc_leg)):
                if name in want:
                    try:
                        out.setdefault("extra", {})[name] = fn(torch, args)
                    except Exception as ex:      # a side leg never takes the headline down
                        out.setdefault("extra", {})[name] = {"error": repr(ex)}
                    import gc
                    gc.collect()
                    torch.cuda.empty_cache()
                    SECTIONS.mark(name)
            if other and other in out and "extra" in out:
                out["extra"]["kappa_out_" + other] = out[other].get("kappa_out")
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(N, args.res)
            SECTIONS.mark("cpu_baseline")
        out["sections_s"] = SECTIONS.t
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
