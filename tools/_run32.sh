#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02aa; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_lensing_gpu.py tests/test_maps_gpu.py tests/test_fullsize_gpu.py -x -q -k "pol or mv or column_grid or config3 or taper or mask or window" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
python tools/config_bench.py all --no-dense 2>&1 | tail -8
