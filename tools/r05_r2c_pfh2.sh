#!/bin/bash
# sweep of the float64 R2C's early-prefetch count PFH (experiment builds -DOA_RS4096_PFH=4 / 8 / 12, OA_RS4096_PF=1) against no prefetch:
# in-step durations (headline and wide-band kernels), then the bench headline twice each
TAG=${1:-r05pfh2}; O=gpurun_out/$TAG; mkdir -p $O
for v in pfh4:pf4 pfh8:pf8 pfh12:pf12 exp:nopf; do
  lib=${v%%:*}; name=${v##*:}
  export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_$lib.so
  if [ $name = nopf ]; then unset OA_RS4096_PF; else export OA_RS4096_PF=1; fi
  bash tools/trace_step.sh $TAG/$name --prec f64 > $O/$name.txt 2>&1
  echo "== $name"; grep -E "row_r2c|kernel sum" $O/$name.txt | cut -c1-200
  bash tools/trace_step.sh $TAG/wb_$name --prec f64 --tlmax 6000 > $O/wb_$name.txt 2>&1
  grep -E "row_r2c" $O/wb_$name.txt | cut -c1-200
done
for rep in 1 2; do for v in exp:nopf pfh8:pf8 pfh12:pf12 pfh4:pf4; do
  lib=${v%%:*}; name=${v##*:}
  export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_$lib.so
  if [ $name = nopf ]; then unset OA_RS4096_PF; else export OA_RS4096_PF=1; fi
  python3 bench.py --prec f64 --also none --no-extras --no-cpu --steps 30 --warmup 5 > $O/bench_${name}_$rep.json 2> $O/bench_$name.err
  echo "bench $name rep$rep: $(grep -o '"value": [0-9.]*' $O/bench_${name}_$rep.json | head -1)"
done; done
