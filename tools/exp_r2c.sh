p() { python tools/r2c_stage_probe.py $1 $2 2>/dev/null | grep stage; }
echo "4096^2: rs2048 f32 (pf) / f32 nopf / f64 (nopf) / f64 pf"; p f32 4096; OA_RS4096_PF=0 p f32 4096; p f64 4096; OA_RS4096_PF=1 p f64 4096
echo "4096^2 general pass: f32 f64"; OA_NO_RS4096=1 p f32 4096; OA_NO_RS4096=1 p f64 4096
echo "8192^2 f32 f64"; p f32 8192; p f64 8192
python -m pytest tests/test_fullsize_gpu.py tests/test_onecall_gpu.py tests/test_lensing_gpu.py -m gpu -x -q -k "full_size or round_trip or config2 or two_maps or binning or mc_run or grids_engaged" 2>&1 | tail -3
