#!/bin/bash
# 16384^2 0.25' (BASELINE config 5): strict parity test + throughput of the R = 8 R-split path, both precisions
TAG=${1:-r04d}; O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_fullsize_gpu.py -x -q -k "16384_strict or 16384_tt_qe" > $O/pytest_16384.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_16384.log
for prec in f64 f32; do
  timeout -k 10 400 python3 bench.py --n 16384 --res 0.25 --no-cpu --no-extras --also none --prec $prec --steps 4 --warmup 2 --batch 16 2> $O/bench_$prec.err > $O/bench_16384_$prec.json
  python3 -c "
import json
d=json.load(open('$O/bench_16384_$prec.json')); r=d['roofline']
print('$prec', round(d['value']), 'recon/s', r['kernel'][:30], r.get('kernel_symbol'), 'frac', round(r['frac'],3), {k[:14]:round(x*1e3,1) for k,x in r['share_of_recon_ms'].items()})"
done
for pf in 0 1; do
  ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_exp.so OA_RS4096_PF=$pf timeout -k 10 400 python3 bench.py --n 16384 --res 0.25 --no-cpu --no-extras --also none --prec f64 --steps 4 --warmup 2 --batch 16 2>> $O/bench_pf.err | python3 -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('f64 PF=$pf', round(d['value']), 'recon/s', {k[:14]:round(x*1e3,1) for k,x in r['share_of_recon_ms'].items()})"
done
