# Dev tool (GPU box): rocprofv3 --kernel-trace --stats of the default bench command -> gpurun_out/quick_stats.{csv,txt}
set -e
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/quick
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- python3 $root/bench.py --steps 25 --warmup 3 --no-cpu-baseline --no-kernel-timing > $out/stats.log 2>&1
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $root/gpurun_out/quick_stats.csv
rm -rf $out/stats
cd $root
python3 tools/kernel_stats_summary.py gpurun_out/quick_stats.csv 400 > gpurun_out/quick_stats.txt
head -3 gpurun_out/quick_stats.txt
