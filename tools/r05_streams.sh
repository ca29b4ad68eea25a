#!/bin/bash
TAG=${1:-r05st}; O=gpurun_out/$TAG; mkdir -p $O
for rep in 1 2; do for prec in f64 f32; do for s in 2 3 4; do
  python3 bench.py --prec $prec --also none --no-extras --no-cpu --steps 30 --warmup 5 --streams $s > $O/${prec}_s${s}_$rep.json 2>/dev/null
  echo "$prec streams=$s rep$rep: $(grep -o '"value": [0-9.]*' $O/${prec}_s${s}_$rep.json | head -1)"
done; done; done
