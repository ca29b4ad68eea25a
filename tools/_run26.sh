#!/bin/bash
export TMPDIR=/tmp
for m in 0 1; do
  export OA_R2C_W64=$m
  echo "OA_R2C_W64=$m"
  timeout -k 5 120 python tools/r2c_bench.py 8192 380 100
done
