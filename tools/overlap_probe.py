"""Do the HBM-bound row R2C of one reconstruction and the coarse-grid launches of another really overlap?

Stage 0 of oa_qe_tt_stage (row R2C of the 8192^2 map) in a loop on stream A, stages 1..5 (every launch behind it) in a
loop on stream B -- each alone, then both together.  Perfect overlap: together = max(alone); none: together = sum.
    python3 tools/overlap_probe.py [f32|f64] [n_iter]      (env OA_W64_WAVES etc. as usual)
"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import bench                                          # noqa: E402
from orphics_amd._lib import check                    # noqa: E402
from orphics_amd.engine import _ptr                   # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
it = int(sys.argv[2]) if len(sys.argv) > 2 else 300
N = 8192
P = bench.build_pipeline(N, 0.5, prec, torch)
q = P["q"]
eng = P["eng"]
norm = P["geom"].area / float(N * N) ** 2
tm = bench.make_maps(P, torch, 1234, 4)
qa, qb = q, q.fork()
for e in (qa, qb):
    e.bind_bins(P["ids"], P["nids"], norm)
ea, eb = qa._bind_bins(), qb._bind_bins()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
d = P["nids"] - 2
mom = [torch.zeros(1, dtype=torch.int64, device=eng.device), torch.zeros(d, dtype=torch.float64, device=eng.device),
       torch.zeros(d, d, dtype=torch.float64, device=eng.device)]
for e in (ea, eb):                                    # fill every work plane of both plans once
    check(e.lib.oa_qe_tt_moments(e.plan, _ptr(tm[0]), *[_ptr(t) for t in mom], None))
torch.cuda.synchronize()


def r2c(n):
    for i in range(n):
        check(ea.lib.oa_qe_tt_stage(ea.plan, 0, _ptr(tm[i & 3]), sa.cuda_stream))


def coarse(n):
    for i in range(n):
        for k in range(1, 6):
            check(eb.lib.oa_qe_tt_stage(eb.plan, k, _ptr(tm[i & 3]), sb.cuda_stream))


def timed(fns):
    for f in fns:
        f(20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in fns:
        f(it)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e6


# interleave the host issue so that both queues stay fed
def both(n):
    for i in range(n):
        check(ea.lib.oa_qe_tt_stage(ea.plan, 0, _ptr(tm[i & 3]), sa.cuda_stream))
        for k in range(1, 6):
            check(eb.lib.oa_qe_tt_stage(eb.plan, k, _ptr(tm[i & 3]), sb.cuda_stream))


for rep in range(2):
    a = timed([r2c])
    b = timed([coarse])
    c = timed([both])
    print("%s: R2C alone %.1f us, coarse alone %.1f us, both streams %.1f us per iteration (sum %.1f, max %.1f)"
          % (prec, a, b, c, a + b, max(a, b)), flush=True)
