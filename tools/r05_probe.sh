#!/bin/bash
# build + run the stand-alone row-stage probe variants on the GPU box:  gpurun -- 'bash tools/r05_probe.sh <tag>'
TAG=${1:-r05p}; O=gpurun_out/$TAG; mkdir -p $O
CXX="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iorphics_amd/csrc"
build() { $CXX $2 -o $O/$1 tools/probes/rq8_probe.hip 2> $O/$1.err || { echo "build $1 failed"; tail -5 $O/$1.err; }; }
build f32_s "-DPREC=float -DGA=3" & build f32_d "-DPREC=float -DGA=3 -DSTAGED=0" & build f64_s "-DPREC=double -DGA=3" & build f64_d "-DPREC=double -DGA=3 -DSTAGED=0" &
build f32_ss "-DPREC=float -DGA=3 -DSTAMPS" & build f64_ss "-DPREC=double -DGA=3 -DSTAMPS" & build f32_s0 "-DPREC=float -DGA=3 -DLAYQ=0" & build f64_s0 "-DPREC=double -DGA=3 -DLAYQ=0" & build f32_d0 "-DPREC=float -DGA=3 -DLAYQ=0 -DSTAGED=0" & build f64_d0 "-DPREC=double -DGA=3 -DLAYQ=0 -DSTAGED=0" & build f64_ds "-DPREC=double -DGA=3 -DSTAGED=0 -DSTAMPS" &
wait
for b in f32_d f32_s f32_d0 f32_s0 f64_d f64_s f64_d0 f64_s0; do $O/$b 1 40; $O/$b 2 40; done 2>&1 | tee $O/times.txt
$O/f32_ss 1 20 2>&1 | tee $O/stamps_f32.txt
$O/f64_ds 1 20 2>&1 | tee $O/stamps_f64.txt
rm -f $O/f32_d0 $O/f64_d0 $O/f64_ds $O/f32_s $O/f32_d $O/f64_s $O/f64_d $O/f32_ss $O/f64_ss $O/f32_s0 $O/f64_s0
