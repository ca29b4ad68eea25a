#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02y
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_onecall_gpu.py tests/test_lensing_gpu.py tests/test_engine_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for ns in 1 3; do
timeout -k 10 200 python bench.py --no-cpu --no-extras --steps 300 --streams $ns > $O/b$ns.json 2> $O/b$ns.err
python -c "
import json; d=json.load(open('$O/b$ns.json')); print('streams', $ns, round(d['value']), round(d['ms_per_step']*1e3,1), {k[:12]:round(v*1e3,1) for k,v in d['roofline']['share_of_recon_ms'].items()})"
done
bash tools/trace_step.sh r02y
