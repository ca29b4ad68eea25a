#!/bin/bash
# Kernel trace of bench.py's timed loop on one stream: per-position durations and the idle gap in front of each kernel.
#   gpurun -- 'bash tools/trace_step.sh <tag> [bench flags]'
set -u
TAG=${1:-rXX}; shift
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
if [ -n "${PAIR:-}" ]; then MODE="--batch 2"; else MODE="--no-pair --batch 1"; fi     # PAIR=1: two maps per call (oa_qe_tt_moments2)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_trace -- python3 bench.py --steps 30 --warmup 5 --no-cpu --no-extras --also none $MODE --streams 1 "$@" > $O/trace_run.json 2> $O/trace.err
python3 - $O <<'PY'
import csv, glob, sys, statistics
O = sys.argv[1]
rows = []
for f in glob.glob(O + '/p_trace/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
MARK = ('bin_final_kernel', 'col_div_sp_bin_kernel')
steps, cur = [], []
for r in rows:
    cur.append(r)
    if any(m in r['Kernel_Name'] for m in MARK):
        steps.append(cur); cur = []
w = {}
for s in steps:
    sig = tuple(r['Kernel_Name'] for r in s); w[sig] = w.get(sig, 0) + len(sig)
sig = max(w, key=w.get)
good = [s for s in steps if tuple(r['Kernel_Name'] for r in s) == sig][-12:]
out = []
tot_d = tot_g = 0
for i in range(len(sig)):
    d = statistics.median((int(s[i]['End_Timestamp']) - int(s[i]['Start_Timestamp'])) / 1e3 for s in good)
    g = statistics.median(((int(s[i]['Start_Timestamp']) - int(s[i - 1]['End_Timestamp'])) / 1e3) for s in good) if i else 0.0
    r = good[0][i]
    n = r['Kernel_Name'].replace('void ', '').replace('oa::', '')
    n = n[:n.find('(')] if '(' in n else n
    out.append('%-64s grid %-18s wg %4s vgpr %3s lds %6s  %7.1f us  gap %5.1f us' % (n[:64], '%sx%sx%s' % (r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z']), r['Workgroup_Size_X'], r['VGPR_Count'], r['LDS_Block_Size'], d, g))
    tot_d += d; tot_g += g
span = statistics.median((int(s[-1]['End_Timestamp']) - int(s[0]['Start_Timestamp'])) / 1e3 for s in good)
out.append('kernel sum %.1f us, gaps %.1f us, step span %.1f us' % (tot_d, tot_g, span))
open(O + '/trace_step.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
PY
rm -rf $O/p_trace
