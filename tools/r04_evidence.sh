#!/bin/bash
# round-4 evidence: full GPU suite, the driver's bench command, 16384^2 / 4096^2 bench lines, smoke
TAG=${1:-r04z}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
[ -n "$SKIP_TESTS" ] || timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=8 > $O/gpu_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/gpu_pytest.log
timeout -k 10 120 python3 __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
T0=$(date +%s); timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"; 
python3 - $O <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + '/bench.json'))
print(d['value'], d['dtype'], d['ms_per_step'], d['roofline']['bound'], d['roofline']['kernel'][:40], round(d['roofline']['frac'], 3), d['roofline']['traffic'])
print('f32', d['f32']['value'], round(d['f32']['roofline']['frac'], 3), d['f32']['roofline']['traffic'])
ex = d['extra']
for k, v in ex.items():
    if not isinstance(v, dict):
        print(k, v); continue
    print(k, {kk: (round(vv, 2) if isinstance(vv, float) else vv) for kk, vv in v.items() if not isinstance(vv, (dict, list, str))})
print('mv', ex.get('mv')); print('mc', ex.get('mc'))
print(d['hbm'].get('measured_streaming_ceiling'), d['cpu_baseline']['value'])
PY
