#!/bin/bash
# HBM traffic of the step's kernels from the L2 memory-side counters (run on the GPU box):
#   tools/pmc_traffic.sh <tag>     -> gpurun_out/traffic_<tag>/{fetch,write}/.../*counter_collection.csv
# Two passes (FETCH_SIZE and WRITE_SIZE do not fit one), counters only (no API tracing), as the MI355X guide prescribes.
set -e
tag=${1:-r02}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/traffic_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/write.log 2>&1
python3 $root/tools/pmc_traffic_summary.py $out > $out/summary.json
cat $out/summary.json | head -c 1500
