#!/bin/bash
# round 5: where the other configurations stand with the new row stage -- MV trace (f64, f32), 16384^2 both precisions, wide band
TAG=${1:-r05c}; O=gpurun_out/$TAG; mkdir -p $O
bash tools/trace_mv.sh $TAG/mv_f64 > $O/mv_f64.txt 2>&1; tail -22 $O/mv_f64.txt
MV_FLAGS="--prec f32" bash tools/trace_mv.sh $TAG/mv_f32 > $O/mv_f32.txt 2>&1; tail -3 $O/mv_f32.txt
for prec in f64 f32; do
  timeout -k 10 400 python3 bench.py --n 16384 --res 0.25 --no-cpu --no-extras --also none --prec $prec --steps 4 --warmup 2 --batch 16 2> $O/bench_$prec.err > $O/bench_16384_$prec.json
  python3 -c "
import json
d=json.load(open('$O/bench_16384_$prec.json')); r=d['roofline']
print('16384 $prec', round(d['value']), 'recon/s', r.get('kernel_symbol'), 'frac', round(r['frac'],3), {k[:14]:round(x*1e3,1) for k,x in r['share_of_recon_ms'].items()})"
done
for prec in f64 f32; do
  timeout -k 10 400 python3 bench.py --tlmax 6000 --no-cpu --no-extras --also none --prec $prec --steps 10 --warmup 3 2> $O/wb_$prec.err > $O/wb_$prec.json
  python3 -c "
import json
d=json.load(open('$O/wb_$prec.json')); r=d['roofline']
print('wideband $prec', round(d['value']), 'recon/s', {k[:14]:round(x*1e3,1) for k,x in r['share_of_recon_ms'].items()})"
done
