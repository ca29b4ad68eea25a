#!/bin/bash
# Kernel stats of the 4096^2 Monte-Carlo loop (config 4) at two batch sizes: average duration per kernel name.
#   gpurun -- 'bash tools/trace_mc.sh <tag>'
set -u
TAG=${1:-rXX}
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
for B in 1 6; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_mc$B -- python3 tools/config_bench.py mc1 --mc-batch $B > $O/mc_run$B.txt 2> $O/mc$B.err
  python3 - $O/p_mc$B $B <<'PY' | tee $O/mc_stats_b$B.txt
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
print("plan option mc_batch = %s" % sys.argv[2])
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:12]:
    print("%-70s calls %6s  avg %8.1f us  total %8.1f ms" % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
done
