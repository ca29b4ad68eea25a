#!/bin/bash
# Round evidence on the GPU box (run through gpurun from the repo root, two calls: each stays under the 20-minute limit):
#   gpurun --timeout 1200 -- 'bash tools/final_evidence.sh r03z profiles'     # per-step profiles + PMC traffic, f64 and f32
#   gpurun --timeout 1200 -- 'bash tools/final_evidence.sh r03z bench'        # the driver's command with the traffic tables in place
# afterwards, in the build container: cp gpurun_out/<TAG>/summary/* profiles/; cp gpurun_out/<TAG>/bench.json profiles/<TAG>_bench.json ...
export TMPDIR=/tmp
TAG=${1:-r03z}; WHAT=${2:-profiles}
mkdir -p gpurun_out/$TAG
if [ "$WHAT" = profiles ]; then
  bash tools/collect_profiles.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1
  cat gpurun_out/$TAG/summary/${TAG}_step_f64.txt gpurun_out/$TAG/summary/${TAG}_step_f32.txt
elif [ "$WHAT" = pmc ]; then
  bash tools/pmc_step.sh ${TAG} --prec f64 > /dev/null 2>&1; cp gpurun_out/$TAG/pmc_step.txt gpurun_out/$TAG/pmc_step_f64.txt
  bash tools/pmc_step.sh ${TAG} --prec f32 > /dev/null 2>&1; cp gpurun_out/$TAG/pmc_step.txt gpurun_out/$TAG/pmc_step_f32.txt
else
  # bench.py reads profiles/traffic_<PROFILE_TAG>_<prec>[_suffix].json: the tables measured in the `profiles` call (copied into
  # profiles/ in the build container) travel with the snapshot
  timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err; echo "bench rc=$?"
  python3 - $TAG <<'PY'
import json, sys
d = json.load(open('gpurun_out/%s/bench.json' % sys.argv[1]))
print(d['value'], d['dtype'], d['ms_per_step'], d['roofline']['bound'], d['roofline']['kernel'][:40], d['roofline']['frac'], d['roofline']['traffic'])
print('f32', d['f32']['value'], d['f32']['roofline']['frac'], d['f32']['roofline']['traffic'])
print({k: (round(v.get('reconstructions_per_s', v.get('sims_per_s', 0)), 1), v.get('max_rel_bandpower_diff')) for k, v in d['extra'].items()})
print(d['hbm'].get('measured_streaming_ceiling'), d['cpu_baseline']['value'])
PY
fi
