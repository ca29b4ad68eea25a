#!/bin/bash
# PMC passes for the fused row stage in the default (active-column) pipeline; summarised per kernel name.
export TMPDIR=/tmp
O=gpurun_out/pmc_rowqe
rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras --also none --no-pair --streams 1 --batch 1 --preroll 0.1 ${PREC:+--prec $PREC} > /dev/null 2> $O/err$i.txt
done
python3 - <<'PY'
import csv, glob, statistics, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_rowqe/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'row_qe' in k or 'col_div' in k or 'col_legs' in k:
            vals[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in vals.items():
    print(k)
    for c, v in sorted(d.items()):
        print('   %-24s %.4g  (n=%d)' % (c, statistics.median(v), len(v)))
PY
rm -rf $O/p1 $O/p2 $O/p3
