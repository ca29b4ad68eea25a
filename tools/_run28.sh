#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02x
mkdir -p $O
for rep in 1 2; do
for m in 0 1; do
  export OA_R2C_W64=$m
  timeout -k 5 120 python tools/r2c_bench.py 8192 380 100 | grep "width=380"
  timeout -k 10 200 python bench.py --no-cpu --no-extras --steps 300 --streams 3 > $O/b$m.json 2> $O/b$m.err
  python -c "
import json; d=json.load(open('$O/b$m.json')); print('w64=$m', round(d['value']), round(d['ms_per_step']*1e3,1), {k[:12]:round(v*1e3,1) for k,v in d['roofline']['share_of_recon_ms'].items()})"
done
done
