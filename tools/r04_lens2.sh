#!/bin/bash
TAG=${1:-r04g}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_onecall_gpu.py tests/test_lensing_gpu.py -x -q -k "windowed or flat_lensing or unbiased or linear_response or get_sim_teb or several_streams" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for prec in f32 f64; do
  timeout -k 10 400 python3 tools/lensloop_bench.py --prec $prec --nsims 10 2> $O/lens_$prec.err | tee -a $O/lensloop.txt
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_lens -- python3 tools/lensloop_bench.py --prec f64 --nsims 6 > $O/lens_prof_run.txt 2> $O/lens_prof.err
python3 - $O/p_lens 10 <<'PY' | tee $O/lensloop_kernel_stats_f64.txt
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
nsim = float(sys.argv[2])
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("kernel time per simulation (10 simulations in the run, set-up kernels included): %.2f ms" % (tot / nsim / 1e6))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:22]:
    print("%-78s calls %6s  avg %8.1f us  per sim %7.3f ms" % (r['Name'][:78], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / nsim / 1e6))
PY
rm -rf $O/p_lens
timeout -k 10 500 python3 - <<'PY' 2>&1 | tee $O/mc_windowed_streams.txt
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
    for ns in (1, 2, 3, 4):
        for mf in (False, True):
            drv = mc.GaussianN0MonteCarlo(q, tot, edges, comm=None, mean_field=mf, window=taper, streams=ns)
            drv.run_local(range(24)); torch.cuda.synchronize()
            t0 = time.perf_counter(); drv.run_local(range(24, 24 + 240)); torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 240
            print("windowed MC 4096^2 %s streams=%d mean_field=%s: %.1f us per sim = %.0f sims/s" % (prec, ns, mf, dt * 1e6, 1 / dt), flush=True)
    del q
PY
