#!/bin/bash
# A/B of library builds on the MV leg (config 3) and the wideband leg
TAG=${1:-r04mv}; VARS=${2:-default}
O=gpurun_out/$TAG; mkdir -p $O
for v in $VARS; do
  if [ "$v" = default ]; then unset ORPHICS_AMD_LIB; else export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_$v.so; fi
  timeout -k 10 500 python3 bench.py --no-cpu --extras mv,wideband --steps 8 --warmup 2 2> $O/$v.err > $O/$v.json || exit 1
  python3 - $O/$v.json $v <<'PY' | tee -a $O/ab.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
e = d["extra"]
print(sys.argv[2], "headline", round(d["value"]), "mv f64", round(e["mv"]["f64"]["mv_reconstructions_per_s"], 1), "f32", round(e["mv"]["f32"]["mv_reconstructions_per_s"], 1),
      "wideband", round(e["wideband"]["reconstructions_per_s"], 1))
PY
done
