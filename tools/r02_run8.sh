#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02h
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q --deselect "tests/test_fullsize_gpu.py::test_config3_mv_f32_vs_f64_and_fused_vs_modular[8192]" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
for rep in 1 2; do
timeout 600 python bench.py --no-cpu --no-extras --steps 100 2>$O/b.err | python -c "
import json,sys
d=json.load(sys.stdin); print(round(d['value']), d['host_issue_ms_per_step'], {k[:12]:round(v,4) for k,v in d['roofline']['share_of_recon_ms'].items()}, d['roofline']['bound'], round(d['roofline']['frac'],3))"
done
tail -3 $O/b.err
timeout 600 python bench.py --no-cpu --extras bandlimited,wideband,fullres_rows --steps 100 2>$O/b2.err | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d['extra'].items(): print(k, round(v['reconstructions_per_s']), v.get('share_of_recon_ms'))"
tail -3 $O/b2.err
