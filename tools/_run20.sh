#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02p
mkdir -p $O
( timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_pytest.log 2>&1; echo "pytest rc=$?" >> $O/gpu_pytest.log )
tail -3 $O/gpu_pytest.log
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -c 600 $O/bench.json
