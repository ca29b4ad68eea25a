#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02f
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
timeout 900 python examples/qe_unbiasedness.py --nsims 200 --side 1024 --res 0.5 --out $O/r02_unbiasedness_1024.txt > $O/unbias_1024.log 2>&1
tail -8 $O/r02_unbiasedness_1024.txt
timeout 1500 python examples/qe_unbiasedness.py --nsims 200 --side 1200 --res 0.5 --out $O/r02_unbiasedness_1200.txt > $O/unbias_1200.log 2>&1
tail -8 $O/r02_unbiasedness_1200.txt; tail -3 $O/unbias_1200.log
