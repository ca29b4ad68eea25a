#!/bin/bash
# A/B of library builds over the whole default bench line (headline + side legs) on the same box.
#   gpurun -- 'bash tools/r04_ab_full.sh <tag> "<variant names ...>"'
TAG=${1:-r04abf}; VARS=${2:-default}
O=gpurun_out/$TAG; mkdir -p $O
for v in $VARS; do
  if [ "$v" = default ]; then unset ORPHICS_AMD_LIB; else export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_$v.so; fi
  timeout -k 10 500 python3 bench.py --no-cpu 2> $O/$v.err > $O/$v.json || exit 1
  python3 - $O/$v.json $v <<'PY' | tee -a $O/ab.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
e = d.get("extra", {})
def val(x):
    return round(x["value"], 1) if isinstance(x, dict) and "value" in x else None
print(sys.argv[2], "headline", round(d["value"]), "f32", round(d.get("f32", {}).get("value", 0)), {k: val(v) for k, v in e.items() if val(v) is not None})
PY
done
