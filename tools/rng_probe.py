import sys, time, os
sys.path.insert(0, '.')
import numpy as np, torch
from orphics_amd.engine import Engine
e = Engine.get(4096, 4096, "f32")
k = e.grf_hc(1, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(50): e.grf_hc(1 + i, 0, out=k)
torch.cuda.synchronize()
print("grf_hc 4096^2 f32: %.1f us" % ((time.perf_counter() - t0) / 50 * 1e6))
x = e.randn(7, 0, shape=(1 << 24,)).double()
print("randn mean %.2e var-1 %.2e skew %.2e kurt-3 %.2e max|x| %.2f nan %d" % (x.mean(), x.var() - 1, (x ** 3).mean(), (x ** 4).mean() - 3, x.abs().max(), int(torch.isnan(x).sum())))
