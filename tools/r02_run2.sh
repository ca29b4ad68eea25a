#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02b
mkdir -p $O
export ROWQE_CASES="8192,380,664,0;8192,380,664,-1;8192,1139,664,-1;8192,0,0,0"
for lib in "" w2 w4; do
  for c in 0 1 2; do
    echo "== lib=${lib:-default} rows_per_wg=${c}"
    if [ -n "$lib" ]; then export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_$lib.so; else unset ORPHICS_AMD_LIB; fi
    if [ "$c" = "0" ]; then unset OA_QE_ROWS_PER_WG; else export OA_QE_ROWS_PER_WG=$c; fi
    timeout 300 python tools/rowqe_bench.py 20 2>&1 | grep -v amdgpu.ids
  done
done > $O/rowqe_variants.txt 2>&1
unset ORPHICS_AMD_LIB OA_QE_ROWS_PER_WG
cat $O/rowqe_variants.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
