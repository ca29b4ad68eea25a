#!/bin/bash
# The unbiasedness evidence of the round: the notebook criterion at a power-of-two geometry with 500
# simulations, and the paired (+kappa / -kappa) linear-response companions that attribute the residuals.
#   gpurun --timeout 1200 -- 'bash tools/unbiasedness_suite.sh r03'
set -u
TAG=${1:-rXX}
O=gpurun_out/$TAG
mkdir -p $O
E=TT,TE,EE,EB,TB
python3 examples/qe_unbiasedness.py --nsims 500 --side 4096 --estimators $E --out $O/${TAG}_unbiasedness_4096_all.txt > $O/a.log 2>&1
python3 examples/qe_unbiasedness.py --nsims 200 --side 4096 --estimators $E --paired --kappa-scale 1.0 --gradient lensed --out $O/${TAG}_paired_4096_s1_lensed.txt > $O/b.log 2>&1
python3 examples/qe_unbiasedness.py --nsims 200 --side 4096 --estimators $E --paired --kappa-scale 0.25 --gradient lensed --out $O/${TAG}_paired_4096_s025_lensed.txt > $O/c.log 2>&1
python3 examples/qe_unbiasedness.py --nsims 200 --side 4096 --estimators $E --paired --kappa-scale 1.0 --gradient unlensed --out $O/${TAG}_paired_4096_s1_unlensed.txt > $O/d.log 2>&1
python3 examples/qe_unbiasedness.py --nsims 200 --side 4096 --estimators $E --paired --kappa-scale 0.25 --gradient unlensed --out $O/${TAG}_paired_4096_s025_unlensed.txt > $O/e.log 2>&1
grep "^# " $O/${TAG}_*.txt | grep -v "bias_b ="
