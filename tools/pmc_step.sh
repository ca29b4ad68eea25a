#!/bin/bash
# SQ counters of every kernel of the default step (separate --pmc passes, no tracing besides --kernel-trace).
#   gpurun -- 'bash tools/pmc_step.sh <tag> [bench flags]'
TAG=${1:-rXX}; shift
export TMPDIR=/tmp
O=gpurun_out/$TAG
rm -rf $O/pmc; mkdir -p $O/pmc
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/pmc/p$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras --also none --no-pair --streams 1 --batch 1 --preroll 0.1 "$@" > /dev/null 2> $O/pmc/err$i.txt
done
python3 - $O <<'PY'
import csv, glob, statistics, collections, sys
O = sys.argv[1]
vals = collections.OrderedDict()
for f in sorted(glob.glob(O + '/pmc/p*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('void ', '').replace('oa::', '')
        k = k[:k.find('(')] if '(' in k else k
        if any(s in k for s in ('row_', 'col_', 'bin_')):
            key = k[:70] + ' grid=' + r['Grid_Size']
            vals.setdefault(key, collections.defaultdict(list))[r['Counter_Name']].append(float(r['Counter_Value']))
out = []
for k, d in vals.items():
    out.append(k)
    for c, v in sorted(d.items()):
        out.append('   %-26s %.5g  (n=%d)' % (c, statistics.median(v), len(v)))
open(O + '/pmc_step.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out[:400]))
PY
rm -rf $O/pmc/p*
