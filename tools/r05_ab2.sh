#!/bin/bash
# A/B on ONE box (experiment build): narrow divergence / column-stage tiles, the packed filter table -- in-step durations, one stream
TAG=${1:-r05ab2}; O=gpurun_out/$TAG; mkdir -p $O
export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_exp.so
for prec in f64 f32; do
  for v in base OA_DIV_NARROW=1 OA_FBAND_NARROW=1 OA_NO_FBAND_TABLE=1; do
    ( [ $v != base ] && export $v; bash tools/trace_step.sh $TAG/${prec}_$v --prec $prec > $O/${prec}_$v.txt 2>&1 )
    echo "== $prec $v"; grep -E "col_fband|col_div|kernel sum" $O/${prec}_$v.txt
  done
done
