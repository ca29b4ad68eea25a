#!/bin/bash
# build + run the col_fband<LR = 1> probe on the GPU box:  gpurun -- 'bash tools/r05_fband_probe.sh <tag>'
TAG=${1:-r05fb}; O=gpurun_out/$TAG; mkdir -p $O
CXX="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iorphics_amd/csrc"
build() { $CXX $2 -o $O/$1 tools/probes/fband_probe.hip 2> $O/$1.err || { echo "build $1 failed"; tail -5 $O/$1.err; }; }
build f64 "-DPREC=double" & build f32 "-DPREC=float" & build f64_s "-DPREC=double -DSTAMPS" & build f32_s "-DPREC=float -DSTAMPS" &
wait
for b in f64 f32; do $O/$b 30 1139; $O/$b 30 1; done 2>&1 | tee $O/times.txt
$O/f64_s 10 1139 2>&1 | tee $O/stamps_f64.txt
$O/f32_s 10 1139 2>&1 | tee $O/stamps_f32.txt
rm -f $O/f64 $O/f32 $O/f64_s $O/f32_s
