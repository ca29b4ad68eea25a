#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02m
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q -k "not config3" 2>&1 | tail -3
run() { timeout 600 python bench.py --no-cpu --no-extras --steps 200 2>$O/b.err | python -c "
import json,sys
d=json.load(sys.stdin); print('$1', round(d['value']), {k[:12]:round(v,4) for k,v in d['roofline']['share_of_recon_ms'].items()})"; }
run sorted-bin; run sorted-bin
