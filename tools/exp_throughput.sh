v() { python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d['value']), 'issue', round(d['host_issue_ms_per_step'],2), 'ms/step', round(d['ms_per_step'],2))" $1 "$2"; }
for P in f32 f64; do
B="python bench.py --n 4096 --prec $P --also none --no-extras --no-cpu"
for S in 2 3; do $B --streams $S > gpurun_out/_a.json 2>/dev/null; v gpurun_out/_a.json "4096^2 $P pair $S streams"; done
OA_NO_RS4096=1 $B > gpurun_out/_a.json 2>/dev/null; v gpurun_out/_a.json "4096^2 $P general R2C"
done
B="python bench.py --prec f64 --also none --no-extras --no-cpu"; $B > gpurun_out/_a.json 2>/dev/null; v gpurun_out/_a.json "8192^2 f64"
B="python bench.py --prec f32 --also none --no-extras --no-cpu"; $B > gpurun_out/_a.json 2>/dev/null; v gpurun_out/_a.json "8192^2 f32"
python -m pytest tests/test_onecall_gpu.py -m gpu -x -q 2>&1 | tail -2
