#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02v
mkdir -p $O
for m in 0 1; do
  export OA_R2C_STREAM=$m
  echo "OA_R2C_STREAM=$m"
  python tools/r2c_bench.py 8192 380 100
  python tools/r2c_bench.py 4096 190 100
  python tools/r2c_bench.py 16384 760 30
  for ns in 1 3; do
  timeout -k 10 200 python bench.py --no-cpu --no-extras --steps 300 --streams $ns > $O/b_$m$ns.json 2> $O/b_$m$ns.err
  python -c "
import json; d=json.load(open('$O/b_$m$ns.json')); print('stream=$m streams', $ns, round(d['value']), round(d['ms_per_step']*1e3,1), {k[:12]:round(v*1e3,1) for k,v in d['roofline']['share_of_recon_ms'].items()})"
  done
done
unset OA_R2C_STREAM
timeout -k 10 600 python -m pytest tests/test_maps_gpu.py tests/test_engine_gpu.py tests/test_lensing_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
