#!/bin/bash
# Round-4 visit G: whole GPU suite, smoke, the default bench line, the reference's own size, kernel stats of the default bench.
set -o pipefail
OUT=gpurun_out/r04g
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.txt 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $OUT/summary.txt; tail -5 $OUT/pytest_gpu.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; rc=$?; echo "smoke rc=$rc" | tee -a $OUT/summary.txt; tail -2 $OUT/smoke.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; rc=$?; echo "bench rc=$rc" | tee -a $OUT/summary.txt; cat $OUT/bench.json
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --bodies 8192 --steps 2000 --warmup 100 > $OUT/bench_n8192.json 2> $OUT/bench_n8192.err; rc=$?; echo "bench8192 rc=$rc" | tee -a $OUT/summary.txt
for i in 1 2 3; do n-bodysimulation_amd/bin/nbody_headless --n 8192 --steps 20000 --sync-each-step | tail -1 >> $OUT/headless_sync_each_step_n8192.txt; done; cat $OUT/headless_sync_each_step_n8192.txt
exit 0
