#!/bin/bash
# build + run the stand-alone row-stage probe variants on the GPU box:  gpurun -- 'bash tools/r05_probe.sh <tag>'
TAG=${1:-r05p}; O=gpurun_out/$TAG; mkdir -p $O
CXX="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iorphics_amd/csrc"
build() { $CXX $2 -o $O/$1 tools/probes/rq8_probe.hip 2> $O/$1.err || { echo "build $1 failed"; tail -5 $O/$1.err; }; }
build f32_a3 "-DPREC=float -DGA=3" & build f32_a3s "-DPREC=float -DGA=3 -DSTAMPS" & build f64_a3 "-DPREC=double -DGA=3" & build f64_a3s "-DPREC=double -DGA=3 -DSTAMPS" &
build f32_a3t "-DPREC=float -DGA=3 -DRQ8_TOUCH" & build f64_a3t "-DPREC=double -DGA=3 -DRQ8_TOUCH" & build f32_a3ts "-DPREC=float -DGA=3 -DRQ8_TOUCH -DSTAMPS" & build f64_a3ts "-DPREC=double -DGA=3 -DRQ8_TOUCH -DSTAMPS" &
wait
for b in f32_a3 f32_a3t f64_a3 f64_a3t; do $O/$b 1 40; $O/$b 2 40; done 2>&1 | tee $O/times.txt
$O/f32_a3s 1 20 2>&1 | tee $O/stamps_f32.txt
$O/f32_a3ts 1 20 2>&1 | tee $O/stamps_f32t.txt
$O/f64_a3s 1 20 2>&1 | tee $O/stamps_f64.txt
$O/f64_a3ts 1 20 2>&1 | tee $O/stamps_f64t.txt
rm -f $O/f32_a3 $O/f32_a3s $O/f64_a3 $O/f64_a3s $O/f32_a3t $O/f64_a3t $O/f32_a3ts $O/f64_a3ts
