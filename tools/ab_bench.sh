# Dev tool (GPU box): A/B of two builds of the library on one box -- bench.py with TG_LIBRARY=<old>, then with the in-tree build.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OLD=$PWD/tartangan_amd/csrc/libtartangan_amd_old.so
for round in 1 2; do
  TG_LIBRARY=$OLD timeout -k 10 400 python bench.py --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/ab_old_$round.json 2> gpurun_out/ab_old_$round.err
  timeout -k 10 400 python bench.py --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/ab_new_$round.json 2> gpurun_out/ab_new_$round.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/ab_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], d['value'])
    except Exception as e:
        print(f, 'unreadable', e)
PY
