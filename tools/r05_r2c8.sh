#!/bin/bash
# A/B on ONE box: the 8-points-per-thread row R2C (default) against the 16-point one (OA_NO_R2C8=1, experiment build): parity, in-step
# kernel durations (one stream), then the bench's headline twice each.
TAG=${1:-r05r2c8}; O=gpurun_out/$TAG; mkdir -p $O
export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_exp.so
timeout -k 10 300 python3 -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "tt_bandpowers_match_numpy_oracle_at_full_size or headline" > $O/parity.log 2>&1 || { tail -20 $O/parity.log; exit 1; }
tail -2 $O/parity.log
for prec in f64 f32; do
  for v in new old; do
    if [ $v = old ]; then export OA_NO_R2C8=1; else unset OA_NO_R2C8; fi
    bash tools/trace_step.sh $TAG/trace_${prec}_$v --prec $prec > $O/trace_${prec}_$v.txt 2>&1
    echo "== $prec $v"; grep -E "r2c|TOTAL|total" $O/trace_${prec}_$v.txt | head -6
  done
done
for rep in 1 2; do
for prec in f64 f32; do
  for v in new old; do
    if [ $v = old ]; then export OA_NO_R2C8=1; else unset OA_NO_R2C8; fi
    python3 bench.py --prec $prec --also none --no-extras --no-cpu --steps 30 --warmup 5 > $O/${prec}_${v}_$rep.json 2> $O/${prec}_${v}_$rep.err
    echo "$prec $v rep$rep: $(grep -o '"value": [0-9.]*' $O/${prec}_${v}_$rep.json | head -1)"
  done
done
done
