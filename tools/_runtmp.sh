#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02ad; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/gpu_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/gpu_pytest.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
