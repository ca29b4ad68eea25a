#!/bin/bash
# job-level A/B (two maps per call, auto streams): narrow (half-width) tiles of the divergence / column-stage launches, experiment build
TAG=${1:-r05nj}; O=gpurun_out/$TAG; mkdir -p $O
export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_exp.so
for rep in 1 2; do for prec in f64 f32; do for v in base OA_DIV_NARROW=1 OA_FBAND_NARROW=1; do
  ( [ $v != base ] && export $v; python3 bench.py --prec $prec --also none --no-extras --no-cpu --steps 30 --warmup 5 > $O/${prec}_${v}_$rep.json 2> $O/err.txt )
  echo "$prec $v rep$rep: $(grep -o '"value": [0-9.]*' $O/${prec}_${v}_$rep.json | head -1)"
done; done; done
