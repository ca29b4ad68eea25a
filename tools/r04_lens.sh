#!/bin/bash
# lens-loop (oa_lens_maps) + windowed MC: tests, throughput, rocprofv3 kernel tables of one simulation
TAG=${1:-r04f}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_onecall_gpu.py tests/test_lensing_gpu.py -x -q -k "windowed or flat_lensing or unbiased or linear_response" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for prec in f32 f64; do
  timeout -k 10 400 python3 tools/lensloop_bench.py --prec $prec --nsims 10 2> $O/lens_$prec.err | tee -a $O/lensloop.txt
done
# kernel table of the lens loop (f64): stats of a 6-simulation run
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_lens -- python3 tools/lensloop_bench.py --prec f64 --nsims 6 > $O/lens_prof_run.txt 2> $O/lens_prof.err
python3 - $O/p_lens 10 <<'PY' | tee $O/lensloop_kernel_stats_f64.txt
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
nsim = float(sys.argv[2])    # 2 warm-up + 6 timed + 2 staged
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("kernel time per simulation (10 simulations in the run, set-up kernels included): %.2f ms" % (tot / nsim / 1e6))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:22]:
    print("%-78s calls %6s  avg %8.1f us  per sim %7.3f ms" % (r['Name'][:78], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / nsim / 1e6))
PY
rm -rf $O/p_lens
# kernel table of the windowed MC loop (f32, 4096^2)
cat > /tmp/winmc.py <<'PY'
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from orphics_amd import cosmology, lensing, maps, mc
from orphics_amd.geometry import FlatGeometry
N, res, prec = 4096, 0.5, sys.argv[1]
shape = (N, N); g = FlatGeometry.from_res(shape, res); th = cosmology.default_theory(); ml = g.modlmap()
beam = maps.gauss_beam(ml, 1.5); noise = np.full(shape, cosmology.white_noise_power(1.0))
tot = (th.lCl("TT", ml) * beam ** 2 + noise)[:, :N // 2 + 1]
edges = np.linspace(20, 3500, 20); taper, w2 = maps.get_taper(shape, g)
q = lensing.qest(shape, g, th, noise2d=noise, beam2d=beam, kmask=((ml > 300) & (ml < 2000)).astype(np.int64), kmask_K=((ml > 20) & (ml < 3500)).astype(np.int64),
                 unlensed_equals_lensed=True, dtype=prec)
drv = mc.GaussianN0MonteCarlo(q, tot, edges, comm=None, mean_field=False, window=taper)
drv.run_local(range(200)); torch.cuda.synchronize()
PY
for prec in f32 f64; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_win -- python3 /tmp/winmc.py $prec > /dev/null 2> $O/win_prof.err
python3 - $O/p_win $prec <<'PY' | tee $O/winmc_kernel_stats_$prec.txt
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
print("windowed MC 4096^2 %s, 200 simulations: per-kernel averages" % sys.argv[2])
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:14]:
    print("%-78s calls %6s  avg %8.1f us  total %8.1f ms" % (r['Name'][:78], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
rm -rf $O/p_win
done
