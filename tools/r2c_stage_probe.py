"""Stage 0 (row R2C of the from-map path) of the bench job alone, back to back: us per launch.
    ORPHICS_AMD_LIB=<variant .so> python3 tools/r2c_stage_probe.py [f32|f64] [N] [res_arcmin]"""
import sys
import time

import torch

sys.path.insert(0, '.')
import bench                                          # noqa: E402
from orphics_amd._lib import check                    # noqa: E402
from orphics_amd.engine import _ptr, _stream          # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
res = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
P = bench.build_pipeline(N, res, prec, torch)
q = P["q"]
tm = bench.make_maps(P, torch, 1234, 4)
q.bind_bins(P["ids"], P["nids"], 1.0)
e = q._bind_bins()
for st in (0,):
    for i in range(20):
        check(e.lib.oa_qe_tt_stage(e.plan, st, _ptr(tm[i & 3]), _stream()))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(200):
        check(e.lib.oa_qe_tt_stage(e.plan, st, _ptr(tm[i & 3]), _stream()))
    torch.cuda.synchronize()
    print("%s N=%d stage %d: %.1f us per launch (rsplit R = %d)" % (prec, N, st, (time.perf_counter() - t0) / 200 * 1e6, e.lib.oa_plan_rsplit(e.plan)), flush=True)
