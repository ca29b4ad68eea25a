#!/bin/bash
# float64 row R2C (row_r2c_rs4096): no prefetch (product) against the next row's taps requested in two halves, PFH of them right after stage 0
# and the rest behind the first sub-transform stage (OA_RS4096_PF=1, experiment builds with -DOA_RS4096_PFH=16 / 8 / 0)
TAG=${1:-r05pfh}; O=gpurun_out/$TAG; mkdir -p $O
for v in exp:nopf exp:pf16 pfh8:pf8 pfh0:pf0; do
  lib=${v%%:*}; name=${v##*:}
  export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_$lib.so
  if [ $name = nopf ]; then unset OA_RS4096_PF; else export OA_RS4096_PF=1; fi
  bash tools/trace_step.sh $TAG/$name --prec f64 > $O/$name.txt 2>&1
  echo "== $name"; grep -E "row_r2c|kernel sum" $O/$name.txt | cut -c1-200
done
for v in exp:nopf pfh8:pf8 pfh0:pf0; do
  lib=${v%%:*}; name=${v##*:}
  export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_$lib.so
  if [ $name = nopf ]; then unset OA_RS4096_PF; else export OA_RS4096_PF=1; fi
  python3 bench.py --prec f64 --also none --no-extras --no-cpu --steps 30 --warmup 5 > $O/bench_$name.json 2> $O/bench_$name.err
  echo "bench $name: $(grep -o '"value": [0-9.]*' $O/bench_$name.json | head -1)"
done
