#!/bin/bash
# Round profile collection on the GPU box (run from the repo root through gpurun):
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r02a [extra bench flags]'
# Per configuration three SEPARATE rocprofv3 runs (kernel trace + stats; FETCH_SIZE; WRITE_SIZE), the program directly
# after `--`, one HIP stream and one realisation per call (--no-pair) so that the dispatch order is the step order of ONE
# reconstruction (tools/summarise_step.py); the headline run issues two realisations per call, whose launches behind the
# row transforms are the same kernels on twice the grid.
set -u
TAG=${1:-rXX}; shift
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O/summary
prof() {   # prof <suffix> <bench flags...>
  local SUF=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_stats$SUF -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-extras --also none --no-pair --streams 1 --batch 1 "$@" > $O/stats_run$SUF.json 2> $O/stats$SUF.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p_fetch$SUF -- python3 bench.py --steps 6 --warmup 1 --no-cpu --no-extras --also none --no-pair --streams 1 --batch 1 --preroll 0.2 "$@" > /dev/null 2> $O/fetch$SUF.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p_write$SUF -- python3 bench.py --steps 6 --warmup 1 --no-cpu --no-extras --also none --no-pair --streams 1 --batch 1 --preroll 0.2 "$@" > /dev/null 2> $O/write$SUF.err
  python3 tools/summarise_step.py $O/p_stats$SUF $O/p_fetch$SUF $O/p_write$SUF $O/summary $TAG $SUF > $O/summary/${TAG}_step$SUF.txt 2>&1
  rm -rf $O/p_stats$SUF $O/p_fetch$SUF $O/p_write$SUF
}
# both precisions: f64 is the headline (the reference's arithmetic), f32 the second block of the bench line
for PREC in ${PRECS:-f64 f32}; do
  prof _$PREC --prec $PREC "$@"
  if [ -z "${ONLY_DEFAULT:-}" ]; then
    prof _${PREC}_fullrows --prec $PREC --row-grid full "$@"
    prof _${PREC}_dense --prec $PREC --no-prune "$@"
    prof _${PREC}_wideband --prec $PREC --tlmax 6000 "$@"      # SURVEY 8(d)'s T filter to ell = 6000: the R = 2 split
  fi
  cat $O/summary/${TAG}_step_$PREC.txt
done
