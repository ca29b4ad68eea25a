#!/bin/bash
# the wide band (T filter ell < 6000) at 8192^2: oracle parity, then the step's kernel table in both precisions and the bench's side leg
TAG=${1:-r05wb}; O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 400 python3 -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "tt_bandpowers_match_numpy_oracle_at_full_size and 6000" > $O/parity.log 2>&1 || { tail -30 $O/parity.log; exit 1; }
tail -3 $O/parity.log
for prec in f64 f32; do
  bash tools/trace_step.sh $TAG/trace_$prec --prec $prec --tlmax 6000 > $O/trace_$prec.txt 2>&1
  cat $O/trace_$prec.txt
done
for prec in f64 f32; do
  python3 bench.py --prec $prec --tlmax 6000 --also none --no-extras --no-cpu --steps 30 --warmup 5 > $O/bench_$prec.json 2> $O/bench_$prec.err
  echo "$prec: $(grep -o '"value": [0-9.]*' $O/bench_$prec.json | head -1)"
done
