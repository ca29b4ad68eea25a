#!/bin/bash
# round 5: the step's kernels inside the bench's timed loop (rocprofv3 kernel trace, one stream), optionally several libraries
#   gpurun -- 'bash tools/r05_rq8.sh <tag> [lib:env ...]'
set -u
TAG=${1:-r05a}; shift
O=gpurun_out/$TAG
mkdir -p $O
run() {  # run <name> <lib or ''> <env> <bench flags>
  local NAME=$1 LIB=$2 ENVV=$3; shift 3
  ( [ -n "$LIB" ] && export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_$LIB.so; [ -n "$ENVV" ] && export $ENVV; bash tools/trace_step.sh $TAG/$NAME "$@" > /dev/null 2>&1 )
  echo "== $NAME"; cat $O/$NAME/trace_step.txt; grep -o '"value": [0-9.]*' $O/$NAME/trace_run.json | head -1
}
run f64 "" "" --prec f64
run f32 "" "" --prec f32
PAIR=1 run f64_pair "" "" --prec f64
PAIR=1 run f32_pair "" "" --prec f32
