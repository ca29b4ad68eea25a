for P in f32 f64; do
echo "== $P default"; bash tools/trace_step.sh r03y_n0 --prec $P | tail -5
echo "== $P fband narrow"; OA_FBAND_NARROW=1 bash tools/trace_step.sh r03y_n1 --prec $P | tail -5 | grep -E "fband|sum"
echo "== $P div narrow"; OA_DIV_NARROW=1 bash tools/trace_step.sh r03y_n2 --prec $P | tail -5 | grep -E "div|sum"
done
python -m pytest tests/test_onecall_gpu.py -m gpu -x -q 2>&1 | tail -2
OA_FBAND_NARROW=1 OA_DIV_NARROW=1 python -m pytest tests/test_onecall_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "binning or full_size or two_maps" 2>&1 | tail -2
