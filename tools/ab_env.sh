#!/bin/bash
# usage: tools/ab_env.sh VAR "v1 v2 ..." [bench flags]  -- A/B one environment variable on the same box
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  env $VAR=$v python bench.py --no-cpu --no-extras "$@" 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']['share_of_recon_ms']
print('$VAR=$v', round(d['value']), {k[:10]:round(x,4) for k,x in r.items()})"
done
