#!/bin/bash
# PMC passes (SQ activity, LDS conflicts, HBM traffic) for the kernels whose names match $KERN (regex), bench flags after the tag:
#   gpurun -- 'KERN="col_fband|row_qe8" bash tools/pmc_kernels.sh tag --prec f64 --tlmax 6000'
export TMPDIR=/tmp
TAG=${1:-pmc}; shift
O=gpurun_out/$TAG
rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras --also none --no-pair --streams 1 --batch 1 --preroll 0.1 "$@" > /dev/null 2> $O/err$i.txt
done
KERN="${KERN:-.}" python3 - $O <<'PY'
import csv, glob, statistics, collections, os, re, sys
O = sys.argv[1]
pat = re.compile(os.environ['KERN'])
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if pat.search(k):
            vals[k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
with open(O + '/summary.txt', 'w') as out:
    for k, d in vals.items():
        print(k, file=out)
        for c, v in sorted(d.items()):
            print('   %-24s %.4g  (n=%d)' % (c, statistics.median(v), len(v)), file=out)
print(open(O + '/summary.txt').read())
PY
rm -rf $O/p?
