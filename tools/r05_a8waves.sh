#!/bin/bash
# float64 row stage on the 4096- / 8192-point grids (A = 8, 16) at 4 (product), 3 and 2 waves per SIMD: in-step durations, wide band and fullres_rows
TAG=${1:-r05a8}; O=gpurun_out/$TAG; mkdir -p $O
for v in base a8w3 a8w2; do
  if [ $v = base ]; then unset ORPHICS_AMD_LIB; else export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_$v.so; fi
  bash tools/trace_step.sh $TAG/wb_$v --prec f64 --tlmax 6000 > $O/wb_$v.txt 2>&1
  bash tools/trace_step.sh $TAG/fr_$v --prec f64 --row-grid full > $O/fr_$v.txt 2>&1
  echo "== $v"; grep -E "row_qe8|kernel sum" $O/wb_$v.txt $O/fr_$v.txt | cut -c1-200
done
