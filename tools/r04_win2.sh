#!/bin/bash
TAG=${1:-r04j}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_onecall_gpu.py tests/test_lensing_gpu.py tests/test_maps_gpu.py -x -q -k "windowed or several_streams or mc_run or mc_driver or mapgen or get_sim_teb or flat_lensing" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 500 python3 - <<'PY' 2>&1 | tee $O/mc_windowed.txt
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from orphics_amd import cosmology, lensing, maps, mc
from orphics_amd.geometry import FlatGeometry
N, res = 4096, 0.5
shape = (N, N); g = FlatGeometry.from_res(shape, res); th = cosmology.default_theory(); ml = g.modlmap()
beam = maps.gauss_beam(ml, 1.5); noise = np.full(shape, cosmology.white_noise_power(1.0))
tot = (th.lCl("TT", ml) * beam ** 2 + noise)[:, :N // 2 + 1]
edges = np.linspace(20, 3500, 20); taper, w2 = maps.get_taper(shape, g)
for prec in ("f32", "f64"):
    q = lensing.qest(shape, g, th, noise2d=noise, beam2d=beam, kmask=((ml > 300) & (ml < 2000)).astype(np.int64), kmask_K=((ml > 20) & (ml < 3500)).astype(np.int64),
                     unlensed_equals_lensed=True, dtype=prec)
    for ns in (1, 3):
        for mf in (False, True):
            drv = mc.GaussianN0MonteCarlo(q, tot, edges, comm=None, mean_field=mf, window=taper, streams=ns)
            drv.run_local(range(24)); torch.cuda.synchronize()
            t0 = time.perf_counter(); drv.run_local(range(24, 24 + 240)); torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 240
            print("windowed MC 4096^2 %s streams=%d mean_field=%s: %.1f us per sim = %.0f sims/s" % (prec, ns, mf, dt * 1e6, 1 / dt), flush=True)
    del q
PY
for prec in f32 f64; do
  timeout -k 10 400 python3 tools/lensloop_bench.py --prec $prec --nsims 10 2> $O/lens_$prec.err | tee -a $O/lensloop.txt
done
