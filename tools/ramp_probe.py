#!/usr/bin/env python3
"""Throughput of the bench step in 50-step chunks from a cold start (clock-ramp / first-run behaviour)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

P = bench.build_pipeline(8192, 0.5, "f32", torch)
q, eng = P["q"], P["eng"]
tm = [eng.irfft(eng.grf_hc(1, i, P["cs"]), scale=1.0 / 8192) for i in range(2)]
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 2
qs = [q] + [q.fork() for _ in range(ns - 1)]
st = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(ns - 1)]
kT = [e.eng.hc() for e in qs]; kk = [e.eng.hc() for e in qs]
norm = P["geom"].area / float(8192 ** 2) ** 2
torch.cuda.synchronize()
t00 = time.perf_counter()
for c in range(40):
    t0 = time.perf_counter()
    for i in range(50):
        j = i % ns
        with torch.cuda.stream(st[j]):
            e = qs[j].eng
            e.rfft(tm[i & 1], out=kT[j])
            qs[j].reconstruct_tt_hc(kT[j], out=kk[j])
            e.bin_power(kk[j], kk[j], norm, P["ids"], P["nids"], herm=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("t=%.2fs  %.1f recon/s" % (t1 - t00, 50 / (t1 - t0)), flush=True)
