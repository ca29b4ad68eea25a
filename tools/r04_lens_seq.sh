#!/bin/bash
# ordered kernel sequence (name, duration, gap to the previous kernel) of the last simulation of the lens loop
TAG=${1:-r04s}; PREC=${2:-f64}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/p_seq -- python3 tools/lensloop_bench.py --prec $PREC --nsims 4 > $O/seq_run_$PREC.txt 2> $O/seq_$PREC.err
python3 - $O/p_seq <<'PY' > $O/lens_seq_$PREC.txt
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = rows[-260:]
prev = None
for r in last:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev) / 1e3 if prev else 0.0
    print("%8.1f us  gap %7.1f  %s  grid %s" % ((e - s) / 1e3, gap, r['Kernel_Name'][:110], r.get('Grid_Size_X', '') + 'x' + r.get('Grid_Size_Y', '')))
    prev = e
PY
rm -rf $O/p_seq
tail -3 $O/seq_run_$PREC.txt
