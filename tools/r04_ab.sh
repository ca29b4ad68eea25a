#!/bin/bash
# A/B of library builds on the same box: headline value + per-kernel live times, both precisions.
#   gpurun -- 'bash tools/r04_ab.sh <tag> "<variant names ...>"'      (variants under orphics_amd/variants/; "default" = the product library)
TAG=${1:-r04ab}; VARS=${2:-default}
O=gpurun_out/$TAG; mkdir -p $O
for v in $VARS; do
  for prec in f64 f32; do
    if [ "$v" = default ]; then unset ORPHICS_AMD_LIB; else export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_$v.so; fi
    timeout -k 10 300 python3 bench.py --no-cpu --no-extras --also none --prec $prec --steps 12 --warmup 3 ${BENCH_FLAGS:-} 2> $O/${v}_$prec.err | python3 -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']['share_of_recon_ms']
print('$v $prec', round(d['value']), 'recon/s; us:', {k[:14]:round(x*1e3,1) for k,x in r.items()}, 'sum', round(sum(r.values())*1e3,1))" | tee -a $O/ab.txt
  done
done
