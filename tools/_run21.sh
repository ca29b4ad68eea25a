#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02r
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_lensing_gpu.py tests/test_onecall_gpu.py tests/test_engine_gpu.py -x -q -k "column_grid or pruning or onecall or one_call or fused or mc_driver" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
timeout -k 10 300 python bench.py --no-cpu --no-extras --steps 200 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02r/bench.json'))
print(d['value'], d['ms_per_step'], {k[:14]:round(v*1e3,1) for k,v in d['roofline']['share_of_recon_ms'].items()})
PY
