#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02ab; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_lensing_gpu.py -x -q -k "every_estimator" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
timeout -k 10 900 python examples/qe_unbiasedness.py --nsims 200 --side 1200 --res 0.5 --estimators TT,TE,EE,EB,TB --out $O/r02_unbiasedness_1200_all.txt > $O/unb.log 2>&1; echo "verifier rc=$?"; tail -8 $O/unb.log
