#!/bin/bash
# Kernel trace of tools/overlap_probe.py: per kernel name, the median duration of launches that ran entirely while a
# row R2C kernel of the other stream was in flight, against launches that did not overlap one at all.
#   gpurun -- 'bash tools/overlap_trace.sh <tag> [f32|f64]'
set -u
TAG=${1:-rXX}; PREC=${2:-f32}
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/p_trace -- python3 tools/overlap_probe.py $PREC 120 > $O/probe.txt 2> $O/probe.err
python3 - $O <<'PY'
import csv, glob, sys, statistics, collections
O = sys.argv[1]
rows = []
for f in glob.glob(O + '/p_trace/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
def short(n):
    n = n.replace('void ', '').replace('oa::', '')
    return n[:n.find('(')] if '(' in n else n
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r['Queue_Id']) for r in rows]
ev.sort()
r2c = [(s, e) for s, e, n, q in ev if 'r2c' in n or 'row_fft_kernel' in n]
import bisect
starts = [s for s, e in r2c]
def cover(s, e):
    """fraction of [s, e] during which some R2C launch was running"""
    i = bisect.bisect_right(starts, e)
    tot = 0
    for a, b in r2c[max(0, i - 4):i]:
        tot += max(0, min(e, b) - max(s, a))
    return tot / max(1, e - s)
inside, outside = collections.defaultdict(list), collections.defaultdict(list)
for s, e, n, q in ev:
    if 'r2c' in n or 'row_fft_kernel' in n:
        continue
    c = cover(s, e)
    if c > 0.9: inside[n].append((e - s) / 1e3)
    elif c < 0.1: outside[n].append((e - s) / 1e3)
out = []
for n in sorted(set(inside) | set(outside)):
    a, b = inside.get(n, []), outside.get(n, [])
    out.append('%-62s under R2C: %7.1f us (n=%4d)   alone: %7.1f us (n=%4d)' % (n[:62], statistics.median(a) if a else -1, len(a), statistics.median(b) if b else -1, len(b)))
# R2C durations with and without anything else in flight
oth = [(s, e) for s, e, n, q in ev if not ('r2c' in n or 'row_fft_kernel' in n)]
ost = [s for s, e in oth]
def ocover(s, e):
    i = bisect.bisect_right(ost, e); tot = 0
    for a, b in oth[max(0, i - 40):i]:
        tot += max(0, min(e, b) - max(s, a))
    return tot / max(1, e - s)
w, wo = [], []
for s, e in r2c:
    c = ocover(s, e)
    (w if c > 0.5 else wo if c < 0.05 else []).append((e - s) / 1e3)
out.append('R2C launches: with coarse kernels in flight %.1f us (n=%d), alone %.1f us (n=%d)' % (statistics.median(w) if w else -1, len(w), statistics.median(wo) if wo else -1, len(wo)))
open(O + '/overlap_trace.txt', 'w').write(open(O + '/probe.txt').read() + '\n'.join(out) + '\n')
print('\n'.join(out))
PY
rm -rf $O/p_trace
