#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02l
mkdir -p $O
run() { timeout 600 python bench.py --no-cpu --no-extras --steps 200 2>$O/b.err | python -c "
import json,sys
d=json.load(sys.stdin); print('$1', round(d['value']), {k[:12]:round(v,4) for k,v in d['roofline']['share_of_recon_ms'].items()})"; }
run rp4-default; run rp4-default
for r in 1 2 8; do export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_rp$r.so; run rp$r; run rp$r; done
unset ORPHICS_AMD_LIB
timeout 900 python -m pytest tests/test_maps_gpu.py tests/test_engine_gpu.py tests/test_onecall_gpu.py tests/test_lensing_gpu.py -m gpu -x -q 2>&1 | tail -3
