v() { python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d['value']), 'issue', round(d['host_issue_ms_per_step'],2), 'ms/step', round(d['ms_per_step'],2))" $1 "$2"; }
for P in f64 f32; do
B="python bench.py --prec $P --also none --no-extras --no-cpu"
OA_RS4096_PERSIST=1 $B > gpurun_out/_a.json 2>/dev/null; v gpurun_out/_a.json "$P persistent"
OA_RS4096_PERSIST=0 $B > gpurun_out/_a.json 2>/dev/null; v gpurun_out/_a.json "$P one wg per group"
OA_RS4096_PERSIST=0 $B --streams 3 > gpurun_out/_a.json 2>/dev/null; v gpurun_out/_a.json "$P one wg per group, 3 streams"
OA_RS4096_PERSIST=0 OA_RS4096_PF=0 $B > gpurun_out/_a.json 2>/dev/null; v gpurun_out/_a.json "$P one wg per group, nopf"
done
OA_RS4096_PERSIST=0 python tools/r2c_stage_probe.py f64 | grep stage
OA_RS4096_PERSIST=0 OA_RS4096_PF=0 python tools/r2c_stage_probe.py f32 | grep stage
