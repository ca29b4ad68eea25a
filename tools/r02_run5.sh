#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02e
mkdir -p $O
python examples/qe_unbiasedness.py --nsims 12 --side 512 --res 1.0 --estimators TT,EB,EE,TE,TB > $O/unbias_512.txt 2>&1
tail -30 $O/unbias_512.txt
export ROWQE_CASES="8192,380,664,-1;8192,1139,664,-1;4096,190,332,-1;16384,760,1328,-1"
for lib in "" p2 p4; do
    echo "== lib=${lib:-default}"
    if [ -n "$lib" ]; then export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_$lib.so; else unset ORPHICS_AMD_LIB; fi
    timeout 300 python tools/rowqe_bench.py 20 2>&1 | grep -v amdgpu.ids
done > $O/rowqe_variants.txt 2>&1
cat $O/rowqe_variants.txt
for lib in "" c6; do
    echo "== lib=${lib:-default}"
    if [ -n "$lib" ]; then export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_$lib.so; else unset ORPHICS_AMD_LIB; fi
    timeout 600 python bench.py --no-cpu --no-extras --steps 100 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print(round(d['value']), {k[:12]:round(v,4) for k,v in d['roofline']['share_of_recon_ms'].items()})"
done
