#!/bin/bash
# 8192^2 MV at float64: kernel table of one oa_qe_mv call; then the single-pass leg kernel on the 2048-row grid (OA_LEGS_SP_2048=1, experiment build)
TAG=${1:-r05mv}; O=gpurun_out/$TAG; mkdir -p $O
MV_FLAGS=--f64 bash tools/trace_mv.sh $TAG/base > $O/base.txt 2>&1; cat $O/base.txt | cut -c1-150
export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_exp.so
python3 tools/config_bench.py mv --no-dense --f64 2>/dev/null | tail -1
OA_LEGS_SP_2048=1 python3 tools/config_bench.py mv --no-dense --f64 2>/dev/null | tail -1
OA_LEGS_SP_2048=1 python3 tools/config_bench.py mv --no-dense 2>/dev/null | tail -1
python3 tools/config_bench.py mv --no-dense 2>/dev/null | tail -1
