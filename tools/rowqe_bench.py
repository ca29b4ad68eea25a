#!/usr/bin/env python3
"""Micro-benchmark of the fused row stage alone (oa_qe_rows) over a few (N, win, wout, mrow) cases, with a
correctness check of every case against the full-length transform.  usage: python tools/rowqe_bench.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from orphics_amd.engine import Engine  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cases = [(8192, 380, 664, 0), (8192, 380, 664, -1), (8192, 1139, 664, 0), (8192, 1139, 664, -1), (8192, 0, 0, 0),
         (4096, 190, 332, -1), (16384, 760, 1328, -1), (16384, 760, 1328, 0)]
if os.environ.get("ROWQE_CASES"):
    cases = [tuple(int(v) for v in c.split(",")) for c in os.environ["ROWQE_CASES"].split(";")]
for (N, win, wout, mrow) in cases:
    e = Engine.get(N, N, "f32")
    W = N // 2 + 1
    wi = win or W
    g = torch.Generator(device="cuda").manual_seed(1)
    ins = []
    for _ in range(3):
        k = e.hc()
        k[:, :wi] = torch.randn(N, wi, dtype=e.cdt, device="cuda", generator=g)
        k[:, 0] = k[:, 0].real.to(e.cdt)
        ins.append(k)
    px, py, rx, ry = e.hc(), e.hc(), e.hc(), e.hc()
    e.qe_rows(ins[0], ins[1], ins[2], rx, ry, scale=1.0, win=win, wout=wout, mrow=0)
    e.qe_rows(ins[0], ins[1], ins[2], px, py, scale=1.0, win=win, wout=wout, mrow=mrow)
    wo = wout or W
    err = float(((px[:, :wo] - rx[:, :wo]).abs().max() / rx[:, :wo].abs().max()).item())
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps):
        e.qe_rows(ins[0], ins[1], ins[2], px, py, scale=1.0, win=win, wout=wout, mrow=mrow)
    b.record(); torch.cuda.synchronize()
    print("N=%5d win=%4d wout=%4d mrow=%5d : %8.1f us   max rel diff vs full-length %.2e" % (N, win, wout, mrow, a.elapsed_time(b) / reps * 1e3, err), flush=True)
    del ins, px, py, rx, ry
    torch.cuda.empty_cache()
