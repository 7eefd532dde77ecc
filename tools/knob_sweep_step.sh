# Dev tool (GPU box): whole-step time under each development knob (one at a time against the defaults), two runs each.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  for r in 1 2; do
    env "$@" python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.4f ms' % d['ms_per_step'])"
  done
}
for kv in X=0 TG_DMA_CK=4 TG_DMA_CK=8 TG_DMA_DB=0 TG_DMA_TILE=512 TG_DMA_WGRAD_DB=1 TG_DMA_PRIO=1 TG_DMA_KSPLIT=0 TG_DMA_S2=0 TG_DMA_WGRAD=0 TG_BN_SMALL_N=8192 TG_BN_SMALL_N=4096 TG_CONV_1X1=0 TG_PREFETCH_RNG=0 X=1; do
  echo "== $kv"
  run $kv
done
