#!/usr/bin/env python3
"""Row R2C pass alone (bench.py's stage 0), at a given active width.  usage: python tools/r2c_bench.py [N] [width] [reps]
ORPHICS_AMD_LIB selects a variant build (tools/build_variant.sh)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from orphics_amd.engine import Engine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
width = int(sys.argv[2]) if len(sys.argv) > 2 else 380
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
e = Engine.get(N, N, "f32")
r1 = torch.randn(N, N, device="cuda", dtype=e.rdt)
s1 = e.hc()
ref = torch.fft.rfft(r1.double(), dim=1)
e.fft_pass(0, r1, s1, width=width)
torch.cuda.synchronize()
err = float((s1[:, :width].to(torch.complex128) - ref[:, :width]).abs().max() / ref[:, :width].abs().max())
print("max rel err of the first %d columns vs torch.fft.rfft (f64): %.3g" % (width, err), flush=True)
del ref
for w in (width, 0):
    for _ in range(5):
        e.fft_pass(0, r1, s1, width=w)
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps):
        e.fft_pass(0, r1, s1, width=w)
    b.record(); torch.cuda.synchronize()
    dt = a.elapsed_time(b) / reps * 1e-3
    wc = w if w else N // 2 + 1
    nb = 4 * N * N + 8 * N * wc
    print("%s row_r2c N=%d width=%d  %.1f us  %.2f TB/s" % (os.environ.get("ORPHICS_AMD_LIB", "default").split("_")[-1], N, wc, dt * 1e6, nb / dt / 1e12), flush=True)
