#!/usr/bin/env python3
"""Does capturing one reconstruction step in a HIP graph (torch.cuda.CUDAGraph) beat eager launches?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from orphics_amd.engine import _ptr, _stream
from orphics_amd._lib import check

P = bench.build_pipeline(8192, 0.5, "f32", torch)
q, eng = P["q"], P["eng"]
N = 8192
tm = [eng.irfft(eng.grf_hc(1, i, P["cs"]), scale=1.0 / N) for i in range(2)]
norm = P["geom"].area / float(N * N) ** 2
nids = P["nids"]; d = nids - 2
ns = 2
qs = [q, q.fork()]
kT = [e.eng.hc() for e in qs]; kk = [e.eng.hc() for e in qs]
_, counts = eng.bin_power(kT[0], kT[0], norm, P["ids"], nids, herm=True)
mom = [(torch.zeros(1, dtype=torch.int64, device="cuda"), torch.zeros(d, dtype=torch.float64, device="cuda"),
        torch.zeros(d, d, dtype=torch.float64, device="cuda")) for _ in range(ns)]


def step(j, m):
    e = qs[j].eng
    e.rfft(tm[m], out=kT[j], width=q.leg_cols, rband=q.leg_rows)
    qs[j].reconstruct_tt_hc(kT[j], out=kk[j])
    sums, _ = e.bin_power(kk[j], kk[j], norm, P["ids"], nids, herm=True, active_cols=q.kappa_cols, active_rows=q.kappa_rows)
    check(e.lib.oa_moments_add_binned(_ptr(sums[1:]), _ptr(counts[1:]), d, _ptr(mom[j][0]), _ptr(mom[j][1]), _ptr(mom[j][2]), _stream()))


streams = [torch.cuda.Stream() for _ in range(ns)]
for j in range(ns):
    with torch.cuda.stream(streams[j]):
        for i in range(20):
            step(j, i & 1)
torch.cuda.synchronize()


def timed(fn, n=400):
    fn(20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


def eager1(n):
    with torch.cuda.stream(streams[0]):
        for i in range(n):
            step(0, i & 1)


def eager2(n):
    for i in range(n):
        with torch.cuda.stream(streams[i % 2]):
            step(i % 2, i & 1)


print("eager 1 stream : %.0f recon/s" % timed(eager1), flush=True)
print("eager 2 streams: %.0f recon/s" % timed(eager2), flush=True)
graphs = []
for j in range(ns):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=streams[j]):
        step(j, 0)
        step(j, 1)
    graphs.append(g)


def graph1(n):
    with torch.cuda.stream(streams[0]):
        for i in range(n // 2):
            graphs[0].replay()


def graph2(n):
    for i in range(n // 2):
        with torch.cuda.stream(streams[i % 2]):
            graphs[i % 2].replay()


print("graph 1 stream : %.0f recon/s" % timed(graph1), flush=True)
print("graph 2 streams: %.0f recon/s" % timed(graph2), flush=True)
print("moment counters:", [int(m[0].item()) for m in mom])
