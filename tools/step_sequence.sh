# Dev tool (GPU box): ordered kernel trace of one replayed step of the default bench -> gpurun_out/step_sequence.txt
set -e
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/seq
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out/trace --output-format csv -- python3 $root/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timing "$@" > $out/trace.log 2>&1
cd $root
python3 tools/step_sequence.py $out/trace > gpurun_out/step_sequence.txt
rm -rf $out/trace
tail -1 gpurun_out/step_sequence.txt
