#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02t
mkdir -p $O
for ns in 1 2 3 4 6; do
  timeout -k 10 200 python bench.py --no-cpu --no-extras --steps 300 --streams $ns > $O/b$ns.json 2> $O/b$ns.err
  python -c "
import json; d=json.load(open('$O/b$ns.json')); print('streams', $ns, round(d['value']), round(d['ms_per_step']*1e3,1), 'host', round(d['host_issue_ms_per_step']*1e3,1))"
done
