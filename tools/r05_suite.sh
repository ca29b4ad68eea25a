#!/bin/bash
# full GPU suite with durations + ticket-order A/B (gpurun -- 'bash tools/r05_suite.sh <tag>')
TAG=${1:-r05s}; O=gpurun_out/$TAG; mkdir -p $O
( export ORPHICS_AMD_LIB=$PWD/orphics_amd/variants/liborphics_amd_acqrel.so; bash tools/trace_step.sh $TAG/acqrel_f64 --prec f64 > /dev/null 2>&1; bash tools/trace_step.sh $TAG/acqrel_f32 --prec f32 > /dev/null 2>&1 )
bash tools/trace_step.sh $TAG/relaxed_f64 --prec f64 > /dev/null 2>&1; bash tools/trace_step.sh $TAG/relaxed_f32 --prec f32 > /dev/null 2>&1
for n in relaxed_f64 acqrel_f64 relaxed_f32 acqrel_f32; do echo "== $n"; grep col_div $O/$n/trace_step.txt; done | tee $O/ticket_order.txt
python3 -m pytest tests -q -m gpu --durations=45 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -60 $O/pytest.log
