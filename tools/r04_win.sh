#!/bin/bash
# fused windowed row pass: parity tests + config-4 windowed throughput (fused vs two-pass), and 16384^2 f32 R2C occupancy variants
TAG=${1:-r04e}; O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_onecall_gpu.py tests/test_lensing_gpu.py -x -q -k "windowed" > $O/pytest_win.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_win.log
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
    for fused in (0, 1):
        q.eng.set_option("win_fused", fused)
        for mf in (False, True):
            drv = mc.GaussianN0MonteCarlo(q, tot, edges, comm=None, mean_field=mf, window=taper)
            drv.run_local(range(12)); torch.cuda.synchronize()
            t0 = time.perf_counter(); drv.run_local(range(12, 212)); torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 200
            print("windowed MC 4096^2 %s fused=%d mean_field=%s: %.1f us per sim = %.0f sims/s" % (prec, fused, mf, dt * 1e6, 1 / dt), flush=True)
    del q
PY
for v in default f32occ4; do
  if [ "$v" = default ]; then unset ORPHICS_AMD_LIB; else export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_$v.so; fi
  timeout -k 10 400 python3 bench.py --n 16384 --res 0.25 --no-cpu --no-extras --also none --prec f32 --steps 4 --warmup 2 --batch 16 2>> $O/bench_f32.err | python3 -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('16384 f32 $v', round(d['value']), 'recon/s', {k[:14]:round(x*1e3,1) for k,x in r['share_of_recon_ms'].items()})" | tee -a $O/f32_variants.txt
done
for pf in 0; do
  ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_exp.so OA_RS4096_PF=$pf timeout -k 10 400 python3 bench.py --n 16384 --res 0.25 --no-cpu --no-extras --also none --prec f32 --steps 4 --warmup 2 --batch 16 2>> $O/bench_f32.err | python3 -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('16384 f32 PF=$pf', round(d['value']), 'recon/s', {k[:14]:round(x*1e3,1) for k,x in r['share_of_recon_ms'].items()})" | tee -a $O/f32_variants.txt
done
ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_exp.so OA_NO_RSPLIT=1 timeout -k 10 400 python3 bench.py --n 16384 --res 0.25 --no-cpu --no-extras --also none --prec f32 --steps 4 --warmup 2 --batch 16 2>> $O/bench_f32.err | python3 -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('16384 f32 NO_RSPLIT (round-3 path)', round(d['value']), 'recon/s', {k[:14]:round(x*1e3,1) for k,x in r['share_of_recon_ms'].items()})" | tee -a $O/f32_variants.txt
