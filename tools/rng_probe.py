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
cs = torch.rand((e.ny, e.kp), dtype=e.rdt, device=e.device)
for name, fn in (("grf_hc with covsqrt", lambda i: e.grf_hc(1 + i, 0, cs, out=k)), ("grf_mix one component", lambda i: e.grf_mix(1 + i, [[cs]], out=[k]))):
    fn(0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(50): fn(i)
    torch.cuda.synchronize()
    print("%s: %.1f us" % (name, (time.perf_counter() - t0) / 50 * 1e6))
a = e.grf_hc(5, 0, cs).clone(); b = e.grf_mix(5, [[cs]])[0]
print("equal:", bool(torch.equal(a[:, :e.nxh + 1], b[:, :e.nxh + 1])))
