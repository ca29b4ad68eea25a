#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02k
mkdir -p $O
timeout 600 python -m pytest tests/test_maps_gpu.py tests/test_engine_gpu.py tests/test_onecall_gpu.py -m gpu -x -q 2>&1 | tail -3
run() { timeout 600 python bench.py --no-cpu --no-extras --steps 200 --streams $1 2>$O/b.err | python -c "
import json,sys
d=json.load(sys.stdin); print('streams=$1', round(d['value']), round(d['host_issue_ms_per_step'],4), {k[:12]:round(v,4) for k,v in d['roofline']['share_of_recon_ms'].items()})"; }
for s in 1 2 3 4; do run $s; run $s; done
