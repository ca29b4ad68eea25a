#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02z; mkdir -p $O
timeout -k 10 300 python bench.py --no-cpu --no-extras > $O/bench2.json 2> $O/bench2.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02z/bench2.json'))
print(d['value'], d['ms_per_step'], d['roofline']['bound'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'])
print({k[:14]:round(v*1e3,1) for k,v in d['roofline']['share_of_recon_ms'].items()})
PY
for m in 0 1; do OA_R2C_W64=$m timeout -k 10 300 python bench.py --no-cpu --no-extras 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('w64=$m', round(d['value']), {k[:14]:round(v*1e3,1) for k,v in d['roofline']['share_of_recon_ms'].items()})"; done
