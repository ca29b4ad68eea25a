#!/bin/bash
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_lensing_gpu.py -x -q -k "mc_driver" 2>&1 | tail -12
python tools/config_bench.py mc 2>&1 | tail -5
