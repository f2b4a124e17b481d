#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench (N=1), rocprofv3 kernel stats of the same bench.
# Usage (from the repo root, through gpurun): bash tools/gpu_round.sh <tag>
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_gpu.txt
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
tail -2 $OUT/smoke.txt
timeout -k 10 400 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
cat $OUT/bench.json
REPO=$PWD
( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -- python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $REPO/$OUT/bench_prof.json 2> $REPO/$OUT/prof.err ); echo "rocprof rc=$?" | tee -a $OUT/summary.txt
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -8
