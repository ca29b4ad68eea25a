#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out; bash tools/collect_profiles.sh r02z > gpurun_out/r02z_collect.log 2>&1
cat gpurun_out/r02z/summary/r02z_step.txt
bash tools/pmc_step.sh r02z > /dev/null 2>&1
# bench.py reads profiles/traffic_r02*.json (PROFILE_TAG): give it the tables just measured
for suf in "" _fullrows _dense; do cp gpurun_out/r02z/summary/traffic_r02z$suf.json profiles/traffic_r02$suf.json; done
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02z/bench.json 2> gpurun_out/r02z/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02z/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['bound'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'])
print({k:(round(v.get('reconstructions_per_s',0)), v.get('max_rel_bandpower_diff')) for k,v in d['extra'].items()})
PY
# afterwards, in the build container: cp gpurun_out/r02z/summary/* profiles/; cp gpurun_out/r02z/bench.json profiles/r02z_bench.json;
#   cp gpurun_out/r02z/pmc_step.txt profiles/r02z_pmc_step.txt; for s in "" _fullrows _dense; do cp profiles/traffic_r02z$s.json profiles/traffic_r02$s.json; done
