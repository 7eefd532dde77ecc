#!/bin/bash
# Round-end evidence in one GPU call (run on the GPU box from the repo root):  bash tools/round_evidence.sh r03
#   profiles/<tag>_hbm_traffic.json        PMC FETCH_SIZE / WRITE_SIZE of the conv family (two counter passes)
#   profiles/<tag>_kernel_stats_*.csv      rocprofv3 --kernel-trace --stats of the default bench command
#   profiles/<tag>_bench_default.json      the default bench line (roofline + cpu_baseline)
#   profiles/<tag>_bench_2rank_gloo.json   python3 bench.py --gpus 2 --share-gpu --backend gloo (self-launching rehearsal)
#   profiles/<tag>_gemm_big.txt            tg_gemm_big at the Newton-Schulz shapes (+ rocprof stats of that run)
#   profiles/<tag>_bench_configs.txt       the other single-GPU configurations
set -e
tag=${1:-r03}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/evidence_$tag
mkdir -p $out
bash $root/tools/pmc_traffic.sh $tag > /dev/null
cp $root/gpurun_out/traffic_$tag/summary.json $root/profiles/${tag}_hbm_traffic.json
cp $root/profiles/${tag}_hbm_traffic.json $out/
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- python3 $root/bench.py --steps 25 --warmup 3 --no-cpu-baseline --no-kernel-timing > $out/stats.log 2>&1
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats_128a3_cnn_b64.csv
rm -rf $out/stats
rocprofv3 --kernel-trace --stats -d $out/gstats --output-format csv -- python3 $root/tools/bench_gemm_big.py > $out/${tag}_gemm_big.txt 2>&1
python3 - <<PY >> $out/${tag}_gemm_big.txt
import csv, glob
f = glob.glob('$out/gstats/**/*kernel_stats.csv', recursive=True)[0]
print('--- rocprofv3 --kernel-trace --stats of the run above (gemm_big_kernel rows)')
for r in csv.DictReader(open(f)):
    if 'gemm_big' in r['Name']:
        print(r['Name'][:90], 'calls', r['Calls'], 'avg_us', round(float(r['AverageNs']) / 1e3, 1))
PY
rm -rf $out/gstats
cd $root
python3 tools/kernel_stats_summary.py $out/${tag}_kernel_stats_128a3_cnn_b64.csv 60 > $out/${tag}_kernel_stats_summary.txt
python3 bench.py > $out/${tag}_bench_default.json 2> $out/bench_default.err
python3 bench.py --gpus 2 --share-gpu --backend gloo --steps 3 --warmup 3 > $out/${tag}_bench_2rank_gloo.json 2> $out/bench_2rank.err || echo "2-rank rehearsal failed" >> $out/bench_2rank.err
bash tools/bench_configs.sh > $out/${tag}_bench_configs.txt 2>&1
head -c 600 $out/${tag}_bench_default.json; echo; head -c 400 $out/${tag}_bench_2rank_gloo.json; echo; head -3 $out/${tag}_kernel_stats_summary.txt; cat $out/${tag}_bench_configs.txt; head -8 $out/${tag}_gemm_big.txt
