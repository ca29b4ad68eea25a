#!/bin/bash
export TMPDIR=/tmp
python tools/r2c_bench.py 8192 380 100
ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_noload.so python tools/r2c_bench.py 8192 380 100
ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_nocomp.so python tools/r2c_bench.py 8192 380 100
