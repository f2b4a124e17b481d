#!/bin/bash
# round-3 visit I: GPU tests, the default bench (+ rocprofv3 kernel stats of the same command), the other configs, a soak,
# PMC passes of the N=262144 launch shape, kernel stats at small N
set -o pipefail
REPO=$PWD
OUT=$REPO/gpurun_out/r03_r
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.txt 2>&1; rc=$?
echo "pytest rc=$rc" | tee -a $OUT/summary.txt; tail -4 $OUT/pytest_gpu.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; rc=$?; echo "smoke rc=$rc" | tee -a $OUT/summary.txt; tail -1 $OUT/smoke.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_n262144.json 2> $OUT/bench.err; rc=$?; echo "bench rc=$rc" | tee -a $OUT/summary.txt
[ $rc -ne 0 ] && exit $rc
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_n262144 -- python3 $REPO/bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/prof.err; rc=$?; echo "rocprof rc=$rc" | tee -a $OUT/summary.txt
[ $rc -ne 0 ] && exit $rc
cd $REPO
timeout -k 10 300 python bench.py --bodies 65536 --steps 1000 --warmup 50 --no-cpu-baseline > $OUT/bench_n65536.json 2>> $OUT/bench.err; rc=$?; echo "bench 65536 rc=$rc" | tee -a $OUT/summary.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --dtype f64 --steps 20 --warmup 3 > $OUT/bench_f64.json 2>> $OUT/bench.err; rc=$?; echo "bench f64 rc=$rc" | tee -a $OUT/summary.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --bodies 1048576 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_n1048576.json 2>> $OUT/bench.err; rc=$?; echo "bench 1M rc=$rc" | tee -a $OUT/summary.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 1500 --warmup 20 --repeats 3 --no-cpu-baseline > $OUT/soak_n262144.json 2>> $OUT/bench.err; rc=$?; echo "soak rc=$rc" | tee -a $OUT/summary.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --bodies 8192 --steps 2000 --warmup 100 --no-cpu-baseline --dt 0.1 --init 0 > $OUT/bench_n8192.json 2>> $OUT/bench.err; rc=$?; echo "bench 8192 rc=$rc" | tee -a $OUT/summary.txt
timeout -k 10 300 python bench.py --bodies 16384 --steps 1000 --warmup 100 --no-cpu-baseline > $OUT/bench_n16384.json 2>> $OUT/bench.err; rc=$?; echo "bench 16384 rc=$rc" | tee -a $OUT/summary.txt
./n-bodysimulation_amd/bin/nbody_headless --n 8192 --steps 2000 --init ref --sync-each-step > $OUT/headless_sync_8192.txt 2>&1; echo "headless sync rc=$?" | tee -a $OUT/summary.txt
./n-bodysimulation_amd/bin/nbody_headless --n 8192 --steps 2000 --init ref > $OUT/headless_queued_8192.txt 2>&1; echo "headless queued rc=$?" | tee -a $OUT/summary.txt
for f in bench_n262144 bench_n65536 bench_f64 bench_n1048576 soak_n262144 bench_n8192 bench_n16384; do python - <<PY
import json
d=json.load(open("$OUT/$f.json"))
print("$f", "ms/step %.4f"%d["ms_per_step"], "value %.4e"%d["value"], "frac %.3f"%d["roofline"]["frac"], "repeats", d["repeats"], "kernel_ms %.4f"%d["roofline"]["kernel_ms_per_step"])
PY
done
tail -1 $OUT/headless_sync_8192.txt | cut -c1-200; tail -1 $OUT/headless_queued_8192.txt | cut -c1-200
find $OUT/prof_n262144 -name "*kernel_stats.csv" | head -1 | xargs -r head -4 | cut -c1-200
exit 0
