#!/bin/bash
TAG=${1:-r04k}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_onecall_gpu.py tests/test_fullsize_gpu.py -x -q -k "binning or ticket or two_maps or mc_run or tt_bandpowers or config4 or raw_pointers" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for prec in f32 f64; do
  rm -rf $O/p
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p -- python3 bench.py --steps 6 --warmup 1 --no-cpu --no-extras --also none --no-pair --streams 1 --batch 1 --preroll 0.2 --prec $prec > /dev/null 2>> $O/err.txt
  python3 - $O/p $prec <<'PY' | tee -a $O/overfetch_after.txt
import csv, glob, statistics, sys
vals = []
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'col_div_sp_bin' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE':
            vals.append(float(r['Counter_Value']))
vals = vals[len(vals) // 2:]
print("%s with tile-major ids / Fnorm: col_div_sp_bin FETCH_SIZE x 2048 = %.1f MB (n=%d)" % (sys.argv[2], statistics.median(vals) * 2048 / 1e6, len(vals)))
PY
done
rm -rf $O/p
bash tools/r04_ab.sh $TAG "default"
