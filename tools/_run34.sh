#!/bin/bash
export TMPDIR=/tmp
for rep in 1 2; do
for lib in default c16; do
  if [ $lib = default ]; then unset ORPHICS_AMD_LIB; else export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_$lib.so; fi
  timeout -k 10 200 python bench.py --no-cpu --no-extras --steps 300 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$lib', round(d['value']), round(d['ms_per_step']*1e3,1), {k[:12]:round(v*1e3,1) for k,v in d['roofline']['share_of_recon_ms'].items()})"
done
done
