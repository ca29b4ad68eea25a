#!/bin/bash
# Where do col_div_sp_bin's extra HBM bytes come from (PMC 50.0 / 78.4 MB against 27 / 54 MB algorithmic)?  FETCH_SIZE of that launch with
# the bin-id loads and / or the normalisation-plane loads compiled out (experiment builds; values are then wrong, traffic is what is read)
TAG=${1:-r04y}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
for prec in f32 f64; do
for v in default noids nofn noidsfn; do
  if [ "$v" = default ]; then unset ORPHICS_AMD_LIB; else export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_$v.so; fi
  rm -rf $O/p
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p -- python3 bench.py --steps 6 --warmup 1 --no-cpu --no-extras --also none --no-pair --streams 1 --batch 1 --preroll 0.2 --prec $prec > /dev/null 2>> $O/err.txt
  python3 - $O/p $v $prec <<'PY' | tee -a $O/overfetch.txt
import csv, glob, statistics, sys
vals = []
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'col_div_sp_bin' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE':
            vals.append(float(r['Counter_Value']))
vals = vals[len(vals) // 2:]
print("%s %-8s col_div_sp_bin FETCH_SIZE x 2048 = %.1f MB (n=%d)" % (sys.argv[3], sys.argv[2], statistics.median(vals) * 2048 / 1e6, len(vals)))
PY
done
done
rm -rf $O/p
