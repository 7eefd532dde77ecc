# Dev tool (GPU box): A/B of two source trees on one box -- ab_old/ (an exported earlier commit, built) against the working tree.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for round in 1 2; do
  (cd ab_old && timeout -k 10 400 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-timing "$@") > gpurun_out/ab_old_$round.json 2> gpurun_out/ab_old_$round.err
  timeout -k 10 400 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-timing "$@" > gpurun_out/ab_new_$round.json 2> gpurun_out/ab_new_$round.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/ab_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], d['value'], d.get('final_losses'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
