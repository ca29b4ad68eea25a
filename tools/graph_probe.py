#!/usr/bin/env python3
"""Does replaying the one-call Monte-Carlo step (oa_qe_tt_moments: 10 launches) from a HIP graph beat eager launches?
usage: python tools/graph_probe.py [streams]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N = 8192
P = bench.build_pipeline(N, 0.5, "f32", torch)
tm = bench.make_maps(P, torch, 1234)
R = bench.Runner(P, torch, tm, ns)
for i in range(4 * ns):
    R.step(i)
torch.cuda.synchronize()


def timed(fn, n=600):
    fn(30)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


def eager(n):
    for i in range(n):
        R.step(i)


print("eager, %d streams : %.0f recon/s" % (ns, timed(eager)), flush=True)
UNROLL = 4      # steps per graph (alternating maps)
graphs = []
for j in range(ns):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=R.streams[j]):
        for u in range(UNROLL):
            R.qs[j].tt_moments(R.tmaps[u & 1], R.mom_n[j], R.mom_S[j], R.mom_C[j])
    graphs.append(g)


def graph(n):
    for i in range(n // UNROLL):
        j = i % ns
        with torch.cuda.stream(R.streams[j]):
            graphs[j].replay()


R.zero()
rate = timed(graph)
print("graph, %d streams : %.0f recon/s (%d steps per graph)" % (ns, rate, UNROLL), flush=True)
print("moment counters:", [int(m.item()) for m in R.mom_n])
