#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02d
mkdir -p $O
export ROWQE_CASES="8192,380,664,0;8192,380,664,-1;8192,1139,664,-1;8192,1139,664,0;4096,190,332,-1;16384,760,1328,-1;16384,760,1328,0"
for lib in "" p2 p4; do
    echo "== lib=${lib:-default}"
    if [ -n "$lib" ]; then export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_$lib.so; else unset ORPHICS_AMD_LIB; fi
    timeout 300 python tools/rowqe_bench.py 20 2>&1 | grep -v amdgpu.ids
done > $O/rowqe_variants.txt 2>&1
unset ORPHICS_AMD_LIB
cat $O/rowqe_variants.txt
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 600 python bench.py --no-cpu --extras fullres_rows,wideband > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02d/bench.json'))
print(d['value'], d['host_issue_ms_per_step'], d['roofline']['share_of_recon_ms'])
print({k:v for k,v in d['roofline'].items() if k in ('bound','kernel','achieved','frac')})
for k,v in d['extra'].items(): print(k, v['reconstructions_per_s'], v.get('share_of_recon_ms'), v.get('max_rel_bandpower_diff'))
PY
