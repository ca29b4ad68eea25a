#!/bin/bash
# Round profile collection on the GPU box (run from the repo root through gpurun):
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r01h'
# Three SEPARATE rocprofv3 runs (kernel trace + stats; FETCH_SIZE; WRITE_SIZE), the program directly after `--`.
set -u
TAG=${1:-rXX}
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
python3 bench.py > $O/${TAG}_bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-extras --streams 1 > $O/stats_run.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras --streams 1 --preroll 0.2 > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras --streams 1 --preroll 0.2 > /dev/null 2> $O/write.err
python3 tools/summarise_profiles.py $O/p_stats $O/p_fetch $O/p_write $O/summary $TAG
cp $O/${TAG}_bench.json $O/summary/
rm -rf $O/p_stats $O/p_fetch $O/p_write
tail -c 600 $O/${TAG}_bench.json
