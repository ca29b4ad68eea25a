#!/bin/bash
# SQ counters of the row R2C stage alone (tools/r2c_stage_probe.py), R-split vs plain: gpurun -- 'bash tools/pmc_r2c.sh <tag>'
TAG=${1:-rXX}
export TMPDIR=/tmp
O=gpurun_out/$TAG
rm -rf $O; mkdir -p $O
for mode in rsplit plain; do
  if [ $mode = plain ]; then export OA_NO_RSPLIT=1; else unset OA_NO_RSPLIT; fi
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INST_CYCLES_VMEM SQ_IFETCH" "SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_EXP_GDS SQ_INSTS_BRANCH"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $O/p_${mode}_$i -- python3 tools/r2c_stage_probe.py > /dev/null 2> $O/err_${mode}_$i.txt
  done
done
python3 - $O <<'PY'
import csv, glob, statistics, collections, sys
O = sys.argv[1]
out = []
for mode in ("rsplit", "plain"):
    vals = collections.defaultdict(list)
    for f in sorted(glob.glob(O + '/p_%s_*/**/*counter_collection.csv' % mode, recursive=True)):
        for r in csv.DictReader(open(f)):
            if 'r2c_w64' in r['Kernel_Name']:
                vals[r['Counter_Name']].append(float(r['Counter_Value']))
    out.append(mode)
    for c, v in sorted(vals.items()):
        out.append('   %-26s %.6g  (n=%d)' % (c, statistics.median(v), len(v)))
open(O + '/pmc_r2c.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
PY
rm -rf $O/p_*
