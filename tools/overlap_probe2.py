"""One stream of back-to-back row R2Cs + K streams of coarse-grid chains (stages 1..5): how many coarse chains does it take
to keep up with the HBM-bound R2C stream when each runs ~3x slower under it?
    python3 tools/overlap_probe2.py [f32|f64] [n_iter]"""
import sys
import time

import torch

sys.path.insert(0, '.')
import bench                                          # noqa: E402
from orphics_amd._lib import check                    # noqa: E402
from orphics_amd.engine import _ptr                   # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
it = int(sys.argv[2]) if len(sys.argv) > 2 else 240
N = 8192
P = bench.build_pipeline(N, 0.5, prec, torch)
q = P["q"]
eng = P["eng"]
norm = P["geom"].area / float(N * N) ** 2
tm = bench.make_maps(P, torch, 1234, 4)
KMAX = 6
qs = [q] + [q.fork() for _ in range(KMAX)]
for e in qs:
    e.bind_bins(P["ids"], P["nids"], norm)
es = [e._bind_bins() for e in qs]
ss = [torch.cuda.Stream() for _ in qs]
d = P["nids"] - 2
mom = [torch.zeros(1, dtype=torch.int64, device=eng.device), torch.zeros(d, dtype=torch.float64, device=eng.device),
       torch.zeros(d, d, dtype=torch.float64, device=eng.device)]
for e in es:
    check(e.lib.oa_qe_tt_moments(e.plan, _ptr(tm[0]), *[_ptr(t) for t in mom], None))
torch.cuda.synchronize()


def run(n, K, with_r2c=True):
    for i in range(n):
        if with_r2c:
            check(es[0].lib.oa_qe_tt_stage(es[0].plan, 0, _ptr(tm[i & 3]), ss[0].cuda_stream))
        if K:
            e, s = es[1 + i % K], ss[1 + i % K]
            for k in range(1, 6):
                check(e.lib.oa_qe_tt_stage(e.plan, k, _ptr(tm[i & 3]), s.cuda_stream))


for K in (0, 1, 2, 3, 4, 6):
    for with_r2c in (True, False):
        if not K and not with_r2c:
            continue
        run(24, K, with_r2c)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(it, K, with_r2c)
        torch.cuda.synchronize()
        print("%s: %d coarse stream(s) %s: %.1f us per iteration" % (prec, K, "+ R2C stream" if with_r2c else "alone       ",
                                                                   (time.perf_counter() - t0) / it * 1e6), flush=True)
