#!/bin/bash
# SQ counters of the row R2C stage alone (tools/r2c_stage_probe.py <prec>): gpurun -- 'bash tools/pmc_stage0.sh <tag> <f32|f64> <kernel name substring>'
TAG=${1:-rXX}; PREC=${2:-f64}; KERN=${3:-r2c}
export TMPDIR=/tmp
O=gpurun_out/$TAG
rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p_$i -- python3 tools/r2c_stage_probe.py $PREC > /dev/null 2> $O/err_$i.txt
done
python3 - $O $KERN <<'PY'
import csv, glob, statistics, collections, sys
O, K = sys.argv[1], sys.argv[2]
vals = collections.defaultdict(list)
names = set()
for f in sorted(glob.glob(O + '/p_*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if K in r['Kernel_Name']:
            names.add(r['Kernel_Name'][:90])
            vals[r['Counter_Name']].append(float(r['Counter_Value']))
out = sorted(names)
for c, v in sorted(vals.items()):
    out.append('   %-26s %.6g  (n=%d)' % (c, statistics.median(v), len(v)))
open(O + '/pmc.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
PY
rm -rf $O/p_*
