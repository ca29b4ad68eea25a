#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02u
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_lensing_gpu.py tests/test_onecall_gpu.py tests/test_engine_gpu.py tests/test_maps_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for lib in default fullbar; do
  if [ $lib = default ]; then unset ORPHICS_AMD_LIB; else export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_$lib.so; fi
  python tools/r2c_bench.py 8192 380 100
  for ns in 1 3; do
  timeout -k 10 200 python bench.py --no-cpu --no-extras --steps 300 --streams $ns > $O/b_$lib$ns.json 2> $O/b_$lib$ns.err
  python -c "
import json; d=json.load(open('$O/b_$lib$ns.json')); print('$lib streams', $ns, round(d['value']), round(d['ms_per_step']*1e3,1), {k[:12]:round(v*1e3,1) for k,v in d['roofline']['share_of_recon_ms'].items()})"
  done
done
