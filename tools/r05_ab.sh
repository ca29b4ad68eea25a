#!/bin/bash
# job-level A/B on ONE box: the bench's headline with the 8-point row stage (default) and the 16-point one (OA_NO_ROWQE8=1, experiment build)
TAG=${1:-r05ab}; O=gpurun_out/$TAG; mkdir -p $O
export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_exp.so
for rep in 1 2; do
for prec in f64 f32; do
  for v in new old; do
    if [ $v = old ]; then export OA_NO_ROWQE8=1; else unset OA_NO_ROWQE8; fi
    python3 bench.py --prec $prec --also none --no-extras --no-cpu --steps 30 --warmup 5 > $O/${prec}_${v}_$rep.json 2> $O/${prec}_${v}_$rep.err
    echo "$prec $v rep$rep: $(grep -o '"value": [0-9.]*' $O/${prec}_${v}_$rep.json | head -1)"
  done
done
done
unset OA_NO_ROWQE8
for s in 1 3; do python3 bench.py --prec f64 --also none --no-extras --no-cpu --steps 30 --warmup 5 --streams $s > $O/f64_streams$s.json 2>/dev/null; echo "f64 streams=$s: $(grep -o '"value": [0-9.]*' $O/f64_streams$s.json | head -1)"; done
