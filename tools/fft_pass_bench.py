#!/usr/bin/env python3
"""Micro-benchmark of the individual FFT pass kernels (and friends) -- used under
rocprofv3 --kernel-trace / --pmc to study one kernel at a time.
usage: python tools/fft_pass_bench.py [N] [reps] [prec]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from orphics_amd.engine import Engine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
prec = sys.argv[3] if len(sys.argv) > 3 else "f32"
e = Engine.get(N, N, prec)
es = 4 if prec == "f32" else 8
A = es * N * N
Ah = 2 * es * N * (N // 2 + 1)
r1 = torch.randn(N, N, device="cuda", dtype=e.rdt)
s1, s2 = e.hc(), e.hc()
e.fft_pass(0, r1, s1)


def t(fn):
    for _ in range(2):
        fn()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


s3, s4, s5 = e.hc(), e.hc(), e.hc()
import numpy as np
from orphics_amd.geometry import FlatGeometry
ly, lx = FlatGeometry.from_res((N, N), 0.5).laxes()
e.set_laxes(ly, lx)
f1 = torch.rand(N, e.kp, device="cuda", dtype=e.rdt); f2 = torch.rand(N, e.kp, device="cuda", dtype=e.rdt)
for name, fn, nb in [("legs_cols", lambda: e.qe_legs_cols(s1, s1, f1, f2, out=(s3, s4, s5)), 5 * Ah),
                     ("cols_div", lambda: e.qe_cols_div(s3, s4, f1, out=s5), 3.5 * Ah),
                     ("qe_rows", lambda: e.qe_rows(s1, s2, s3, s4, s5), 5 * Ah), ("row_r2c", lambda: e.fft_pass(0, r1, s1), A + Ah), ("col_pass1", lambda: e.fft_pass(1, s1, s2), 2 * Ah),
                     ("col_pass2", lambda: e.fft_pass(2, s1, s2), 2 * Ah), ("row_c2r", lambda: e.fft_pass(3, s1, r1), A + Ah)]:
    dt = t(fn)
    print("%-10s N=%d %s  %.1f us  %.0f GB/s (algorithmic)" % (name, N, prec, dt * 1e6, nb / dt / 1e9))
