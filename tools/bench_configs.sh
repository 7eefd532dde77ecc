#!/bin/bash
# Dev tool (GPU box): throughput of the other SA-GAN configurations with the current build.
for spec in "128:3 iqn 64" "128 cnn 64" "64:1 cnn 64" "64:1 iqn 64" "64 cnn 64" "32 cnn 64" "256:3 cnn 64" "256:3 iqn 32"; do
  set -- $spec
  python bench.py --config $1 --trainer $2 --batch $3 --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1 $2 b$3', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')"
done
