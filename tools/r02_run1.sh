#!/bin/bash
# first GPU pass of round 2: parity suite, new bench line, per-step profiles
export TMPDIR=/tmp
O=gpurun_out/r02a
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -c 1500 $O/bench.err
timeout 1500 bash tools/collect_profiles.sh r02a > $O/collect.log 2>&1
tail -30 $O/collect.log
