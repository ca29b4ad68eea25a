#!/bin/bash
# 16384^2 0.25' float32 on the R = 8 split (this round) against round 3's kernels (OA_NO_RS8_F32=1, experiment build): parity, then throughput
TAG=${1:-r05c5}; O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "16384_strict or 16384_tt_qe or (tt_bandpowers and 16384)" > $O/pytest_16384.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_16384.log
export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_exp.so
for v in new old; do
  if [ $v = old ]; then export OA_NO_RS8_F32=1; else unset OA_NO_RS8_F32; fi
  timeout -k 10 400 python3 bench.py --n 16384 --res 0.25 --no-cpu --no-extras --also none --prec f32 --steps 4 --warmup 2 --batch 16 2> $O/bench_f32_$v.err > $O/bench_16384_f32_$v.json
  python3 -c "
import json
d=json.load(open('$O/bench_16384_f32_$v.json')); r=d['roofline']
print('f32 $v', round(d['value']), 'recon/s', r.get('kernel_symbol'), 'frac', round(r['frac'],3), {k[:14]:round(x*1e3,1) for k,x in r['share_of_recon_ms'].items()})"
done
