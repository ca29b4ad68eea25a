#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02ac; mkdir -p $O
timeout -k 10 400 python bench.py --n 4096 --no-cpu --extras fullres_rows > $O/bench_4096.json 2> $O/b4096.err; echo "rc=$?"
timeout -k 10 600 python bench.py --n 16384 --res 0.25 --no-cpu --extras fullres_rows --steps 100 > $O/bench_16384.json 2> $O/b16384.err; echo "rc=$?"
python - <<'PY'
import json
for n in (4096, 16384):
    d=json.load(open('gpurun_out/r02ac/bench_%d.json' % n))
    r=d['roofline']
    print(n, round(d['value']), round(d['ms_per_step']*1e3,1), r['bound'], r['kernel'][:30], round(r['frac'],3), {k[:12]:round(v*1e3,1) for k,v in r['share_of_recon_ms'].items()}, r['col_grid']['rows'], r['row_grid']['points'], {k:round(v['reconstructions_per_s']) for k,v in d.get('extra',{}).items()})
PY
