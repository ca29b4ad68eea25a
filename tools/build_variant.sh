#!/bin/bash
# usage: tools/build_variant.sh NAME "<extra hipcc flags>"  -> orphics_amd/variants/liborphics_amd_NAME.so
# (tuning builds selected at run time with ORPHICS_AMD_LIB=<path>; only fft.hip / fft_legs.hip see the flags)
set -e
NAME=$1; FLAGS=$2
cd "$(dirname "$0")/../orphics_amd/csrc"
mkdir -p build_$NAME ../variants
CXX="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -DOA_EXPERIMENTS"
for f in fft fft_legs pipeline; do $CXX $FLAGS -c $f.hip -o build_$NAME/$f.o & done
wait
OBJS="build_$NAME/fft.o build_$NAME/fft_legs.o build_$NAME/pipeline.o"
for f in plan czt elementwise bin rng; do OBJS="$OBJS build/$f.o"; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../variants/liborphics_amd_$NAME.so $OBJS
echo built ../variants/liborphics_amd_$NAME.so
