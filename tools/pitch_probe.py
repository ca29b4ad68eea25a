#!/usr/bin/env python3
"""Does the row pitch of a narrow active-column plane limit the column passes?  Same 8192-point column transforms
of the same 380 / 664 columns on planes of pitch 4112 (nx = 8192) and 528 / 1040 (nx = 1024 / 2048)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orphics_amd.engine import Engine

def t(fn, reps=30):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

for nx, w in ((8192, 380), (1024, 380), (8192, 664), (2048, 664)):
    e = Engine(8192, nx, "f32")
    s1, s2 = e.hc(), e.hc()
    s1.normal_()
    mb = 2 * 8192 * w * 8 / 1e6
    for pid, name in ((1, "pass1"), (2, "pass2 (in place)")):
        us = t(lambda: e.fft_pass(pid, s1, s2, w))
        print("ny=8192 nx=%5d pitch=%5d width=%4d %-16s %7.1f us  %.2f TB/s" % (nx, e.kp, w, name, us, mb / us), flush=True)
    del e, s1, s2
