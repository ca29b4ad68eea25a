#!/bin/bash
# Kernel trace of one 8192^2 MV reconstruction (config 3): kernels of the LAST oa_qe_mv call with durations and the idle
# gap in front of each.   gpurun -- 'bash tools/trace_mv.sh <tag>'
set -u
TAG=${1:-rXX}
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_mv -- python3 tools/config_bench.py mv --no-dense ${MV_FLAGS:-} > $O/mv_run.txt 2> $O/mv.err
python3 - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
rows = []
for f in glob.glob(O + '/p_mv/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# one MV reconstruction ends with its 5th divergence launch; take the last complete run of kernels between zero fills
# one MV reconstruction ends with the sum of the estimators' planes (oa_qe_mv) -- or, per estimator, with its divergence
end = [i for i, r in enumerate(rows) if 'sum_region' in r['Kernel_Name']]
if len(end) >= 2:
    last, start = end[-1], end[-2] + 1
else:
    div = [i for i, r in enumerate(rows) if 'col_div' in r['Kernel_Name']]
    last, start = div[-1], div[-6] + 1
seg = rows[start:last + 1]
out = []
prev_end = int(rows[start - 1]['End_Timestamp'])
tot = gap_tot = 0
for r in seg:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    g = (int(r['Start_Timestamp']) - prev_end) / 1e3
    prev_end = int(r['End_Timestamp'])
    tot += d; gap_tot += max(g, 0)
    out.append("%-90s %7.1f us  gap %6.1f us  grid %s" % (r['Kernel_Name'][:90], d, g, r.get('Grid_Size', '?')))
out.append("kernels %d, busy %.1f us, gaps %.1f us" % (len(seg), tot, gap_tot))
open(O + '/mv_trace.txt', 'w').write("\n".join(out) + "\n")
print("\n".join(out))
PY
tail -2 $O/mv_run.txt
