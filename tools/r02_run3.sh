#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c
mkdir -p $O
export ROWQE_CASES="8192,380,664,0;8192,380,664,-1;8192,1139,664,-1;8192,1139,664,0;8192,0,0,0;4096,190,332,-1;16384,760,1328,-1"
for lib in "" w2 w4; do
  for c in 0 1; do
    echo "== lib=${lib:-default} rows_per_wg=${c}"
    if [ -n "$lib" ]; then export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_$lib.so; else unset ORPHICS_AMD_LIB; fi
    if [ "$c" = "0" ]; then unset OA_QE_ROWS_PER_WG; else export OA_QE_ROWS_PER_WG=$c; fi
    timeout 300 python tools/rowqe_bench.py 20 2>&1 | grep -v amdgpu.ids
  done
done > $O/rowqe_variants.txt 2>&1
unset ORPHICS_AMD_LIB OA_QE_ROWS_PER_WG
cat $O/rowqe_variants.txt
timeout 600 python bench.py --no-cpu --extras fullres_rows,wideband > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02c/bench.json'))
print(d['value'], d['roofline']['share_of_recon_ms'])
for k,v in d['extra'].items(): print(k, v['reconstructions_per_s'], v.get('share_of_recon_ms'))
PY
