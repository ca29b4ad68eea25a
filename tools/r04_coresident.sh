#!/bin/bash
# Experiment: R2C on fewer resident workgroups per CU + narrow (64 KB) coarse tiles, so that the coarse-grid launches of one stream can be
# co-resident with the row R2C of the other (experiment build: switches read from the environment)
TAG=${1:-r04x}; O=gpurun_out/$TAG; mkdir -p $O
export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_exp.so
run() {  # label, env...
  local label=$1; shift
  for prec in f64 f32; do
    env "$@" python3 bench.py --no-cpu --no-extras --also none --prec $prec --steps 12 --warmup 3 ${BENCH_FLAGS:-} 2>> $O/err.txt | python3 -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']['share_of_recon_ms']
print('$label $prec', round(d['value']), 'recon/s; us:', {k[:10]:round(x*1e3,1) for k,x in r.items()})" | tee -a $O/coresident.txt
  done
}
run "default" OA_DUMMY=1
run "narrow tiles" OA_FBAND_NARROW=1 OA_DIV_NARROW=1
run "R2C 1/CU" OA_RS4096_WGS=2
run "R2C 1/CU + narrow" OA_RS4096_WGS=2 OA_FBAND_NARROW=1 OA_DIV_NARROW=1
run "R2C 1.5/CU + narrow" OA_RS4096_WGS=3 OA_FBAND_NARROW=1 OA_DIV_NARROW=1
BENCH_FLAGS="--streams 3" run "3 streams R2C 1/CU + narrow" OA_RS4096_WGS=2 OA_FBAND_NARROW=1 OA_DIV_NARROW=1
BENCH_FLAGS="--streams 3 --no-pair" run "3 streams no-pair R2C 1/CU + narrow" OA_RS4096_WGS=2 OA_FBAND_NARROW=1 OA_DIV_NARROW=1
BENCH_FLAGS="--no-pair" run "2 streams no-pair R2C 1/CU + narrow" OA_RS4096_WGS=2 OA_FBAND_NARROW=1 OA_DIV_NARROW=1
run "R2C 2/CU (f32 only differs) + narrow" OA_RS4096_WGS=4 OA_FBAND_NARROW=1 OA_DIV_NARROW=1
