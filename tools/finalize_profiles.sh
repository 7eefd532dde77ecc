#!/bin/bash
# Round-end evidence in one GPU call (run on the GPU box from the repo root):  bash tools/finalize_profiles.sh r02
#   1. HBM traffic of the step's kernels (two counter passes)   -> gpurun_out/traffic_<tag>/summary.json
#   2. rocprofv3 --kernel-trace --stats of the default bench     -> gpurun_out/stats_<tag>/.../kernel_stats.csv
#   3. in-kernel clock / phase stamps of the conv kernel (diag)  -> gpurun_out/clock_probe.json
#   4. the default bench line                                    -> gpurun_out/bench_<tag>.json
set -e
tag=${1:-r02}
root=$GRAFT_REPO_ROOT
bash $root/tools/pmc_traffic.sh $tag > /dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/stats_$tag --output-format csv -- python3 $root/bench.py --steps 25 --warmup 3 --no-cpu-baseline > $root/gpurun_out/stats_$tag.log 2>&1
cd $root
f=$(find gpurun_out/stats_$tag -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/kernel_stats_$tag.csv
python3 tools/kernel_stats_summary.py gpurun_out/kernel_stats_$tag.csv 12
if [ -f tartangan_amd/csrc/libtartangan_amd_diag.so ]; then
  TG_LIBRARY=$root/tartangan_amd/csrc/libtartangan_amd_diag.so python3 tools/clock_probe.py 2.5 > gpurun_out/clock_probe_$tag.txt 2>&1
fi
