#!/bin/bash
# lens-loop device pipeline (oa_grf_mix + oa_lens_maps_hc): tests, loop rate both precisions, ordered kernel sequence of one simulation
TAG=${1:-r04t}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_lensing_gpu.py -x -q -k "draw_hc or lens_many_hc or get_sim_teb or flat_lensing or lens_many or unbiased or lensed_sims_loop" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for prec in f32 f64; do
  timeout -k 10 400 python3 tools/lensloop_bench.py --prec $prec --nsims 10 2> $O/lens_$prec.err | tee -a $O/lensloop.txt || exit 1
done
bash tools/r04_lens_seq.sh $TAG f64 && bash tools/r04_lens_seq.sh $TAG f32
