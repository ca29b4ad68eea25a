#!/bin/bash
# in-step kernel durations (rocprofv3 kernel trace of the bench's timed loop, two streams x two maps per call as the headline runs) for library variants
TAG=${1:-r04is}; VARS=${2:-default}; PREC=${3:-f64}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
for v in $VARS; do
  if [ "$v" = default ]; then unset ORPHICS_AMD_LIB; else export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$v -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-extras --also none --prec $PREC > $O/run_$v.json 2> $O/err_$v.txt
  python3 - $O/p_$v $v $O/run_$v.json <<'PY' | tee -a $O/instep.txt
import csv, glob, json, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print(sys.argv[2], "value", round(d["value"]))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:5]:
    print("   %-60s calls %5s avg %7.1f us" % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  rm -rf $O/p_$v
done
