#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02g
mkdir -p $O
for lib in "" nt; do
    echo "== lib=${lib:-default}"
    if [ -n "$lib" ]; then export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_$lib.so; else unset ORPHICS_AMD_LIB; fi
    for rep in 1 2; do
    timeout 600 python bench.py --no-cpu --no-extras --steps 100 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print(round(d['value']), {k[:12]:round(v,4) for k,v in d['roofline']['share_of_recon_ms'].items()})"
    done
done 2>&1 | tee $O/nt_ab.txt
unset ORPHICS_AMD_LIB
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 600 python bench.py --n 4096 --no-cpu > $O/bench_4096.json 2>> $O/bench.err; echo "bench4096 rc=$?"
timeout 900 python bench.py --n 16384 --res 0.25 --no-cpu --extras fullres_rows,dense > $O/bench_16384.json 2>> $O/bench.err; echo "bench16384 rc=$?"
timeout 1800 bash tools/collect_profiles.sh r02g > $O/collect.log 2>&1
cat $O/summary/r02g_step.txt $O/summary/r02g_step_fullrows.txt $O/summary/r02g_step_dense.txt
timeout 900 python tools/config_bench.py mv --no-dense > $O/configs.txt 2>&1; timeout 600 python tools/mc_config4.py 1000 >> $O/configs.txt 2>&1; cat $O/configs.txt | grep -v amdgpu
timeout 1500 python examples/qe_unbiasedness.py --nsims 500 --side 1200 --res 0.5 --out $O/r02_unbiasedness_1200_500sims.txt > $O/unbias_1200.log 2>&1
tail -4 $O/r02_unbiasedness_1200_500sims.txt
