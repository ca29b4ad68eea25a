#!/bin/bash
# round 5: the 8-points-per-thread row stage against the 16-point one, inside the bench's step (rocprofv3 kernel trace, one stream)
#   gpurun -- 'bash tools/r05_rq8.sh <tag>'
set -u
TAG=${1:-r05a}
O=gpurun_out/$TAG
mkdir -p $O
python3 -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "tt_bandpowers_match_numpy_oracle_at_full_size and (2048 or 4096)" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
run() {  # run <name> <lib or ''> <env> <bench flags>
  local NAME=$1 LIB=$2 ENVV=$3; shift 3
  ( [ -n "$LIB" ] && export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_$LIB.so; [ -n "$ENVV" ] && export $ENVV; bash tools/trace_step.sh $TAG/$NAME "$@" > /dev/null 2>&1 )
  echo "== $NAME"; cat $O/$NAME/trace_step.txt; grep -o '"value": [0-9.]*' $O/$NAME/trace_run.json | head -1
}
run f64_new "" "" --prec f64
run f32_new "" "" --prec f32
run f64_w3 w3 "" --prec f64

run f64_old w3 OA_NO_ROWQE8=1 --prec f64
run f32_old w3 OA_NO_ROWQE8=1 --prec f32
