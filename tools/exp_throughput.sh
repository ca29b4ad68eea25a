B="python bench.py --prec f32 --also none --no-extras --no-cpu"
v() { python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d['value']), 'issue', round(d['host_issue_ms_per_step'],2), 'ms/step', round(d['ms_per_step'],2))" $1 "$2"; }
for L in occ2 occ1; do
ORPHICS_AMD_LIB=orphics_amd/variants/liborphics_amd_$L.so $B > gpurun_out/_a.json 2>/dev/null; v gpurun_out/_a.json "f32 $L streams3"
ORPHICS_AMD_LIB=orphics_amd/variants/liborphics_amd_$L.so $B --streams 2 > gpurun_out/_a.json 2>/dev/null; v gpurun_out/_a.json "f32 $L streams2"
ORPHICS_AMD_LIB=orphics_amd/variants/liborphics_amd_$L.so $B --streams 4 > gpurun_out/_a.json 2>/dev/null; v gpurun_out/_a.json "f32 $L streams4"
done
$B > gpurun_out/_a.json 2>/dev/null; v gpurun_out/_a.json "f32 default streams3"
