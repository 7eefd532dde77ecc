set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -k "attention" -x > gpurun_out/attn_tests.log 2>&1 || { tail -30 gpurun_out/attn_tests.log; exit 1; }
tail -3 gpurun_out/attn_tests.log
echo OLD; TG_LIBRARY=$PWD/tartangan_amd/csrc/libtartangan_amd_old.so timeout -k 10 300 python tools/bench_attn.py 2>&1 | tee gpurun_out/attn_old.txt
echo NEW; timeout -k 10 300 python tools/bench_attn.py 2>&1 | tee gpurun_out/attn_new.txt
