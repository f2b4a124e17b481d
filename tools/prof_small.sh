#!/bin/bash
set -o pipefail
REPO=$PWD; OUT=$REPO/gpurun_out/small; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for n in 1024 8192 16384 32768; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n$n -- python3 $REPO/bench.py --no-cpu-baseline --bodies $n --steps 500 --warmup 50 > $OUT/n$n.json 2>$OUT/n$n.err
  f=$(find $OUT/n$n -name "*kernel_stats.csv" | head -1); echo "== N=$n"; head -3 $f | cut -d, -f1-4 | cut -c1-150
  python3 -c "import json; d=json.load(open('$OUT/n$n.json')); print('us/step', round(d['ms_per_step']*1e3,2), d['config']['launch'])"
done
