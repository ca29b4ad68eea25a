v() { python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d['value']), 'issue', round(d['host_issue_ms_per_step'],2), 'ms/step', round(d['ms_per_step'],2))" $1 "$2"; }
for P in f64 f32; do
B="python bench.py --prec $P --also none --no-extras --no-cpu"
$B > gpurun_out/_a.json 2>/dev/null; v gpurun_out/_a.json "$P pair 2 streams"
for S in 2 3 4 6; do $B --no-pair --streams $S > gpurun_out/_a.json 2>/dev/null; v gpurun_out/_a.json "$P nopair $S streams"; done
done
