#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench (N=1), rocprofv3 kernel stats of the same bench.
# Usage (from the repo root, through gpurun): bash tools/gpu_round.sh <tag>
# Steps are chained: a step that fails or times out stops the visit (no GPU step after a killed one).
set -o pipefail
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.txt 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $OUT/summary.txt; tail -5 $OUT/pytest_gpu.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; rc=$?; echo "smoke rc=$rc" | tee -a $OUT/summary.txt; tail -2 $OUT/smoke.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; rc=$?; echo "bench rc=$rc" | tee -a $OUT/summary.txt; cat $OUT/bench.json
[ $rc -eq 0 ] || exit $rc
( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -- python3 $REPO/bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-general-path > $REPO/$OUT/bench_prof.json 2> $REPO/$OUT/prof.err ); rc=$?; echo "rocprof rc=$rc" | tee -a $OUT/summary.txt
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -8
exit $rc
