#!/bin/bash
# usage: tools/build_variant.sh NAME "<extra hipcc flags>"  -> orphics_amd/variants/liborphics_amd_NAME.so
# (tuning builds selected at run time with ORPHICS_AMD_LIB=<path>).  Every object of the variant is compiled with the same
# flags (-DOA_EXPERIMENTS + the extra ones) into its own build directory, so no inline function has two definitions in one library.
set -e
NAME=$1; FLAGS=$2
cd "$(dirname "$0")/../orphics_amd/csrc"
mkdir -p ../variants
make -j${JOBS:-8} BUILD=build_$NAME TARGET=../variants/liborphics_amd_$NAME.so EXTRA="-DOA_EXPERIMENTS $FLAGS" > /dev/null
echo built ../variants/liborphics_amd_$NAME.so
