#!/bin/bash
# round-3 visit J: f64 square A/B, PMC passes at N=262144 (square kernel), kernel stats + PMC at N=8192 / 16384 (balanced runs)
set -o pipefail
REPO=$PWD
OUT=$REPO/gpurun_out/r03_j
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 200 ./build/f64bench 262144 5 > $OUT/f64bench.txt 2>&1; rc=$?; echo "f64bench rc=$rc" | tee -a $OUT/summary.txt; cat $OUT/f64bench.txt
[ $rc -ne 0 ] && exit $rc
bash tools/pmc.sh r03_j/pmc262144 > $OUT/pmc262144.log 2>&1; rc=$?; echo "pmc 262144 rc=$rc" | tee -a $OUT/summary.txt
[ $rc -ne 0 ] && { tail -5 $OUT/pmc262144.log; exit $rc; }
bash tools/prof_small.sh r03_j/small 8192 16384 32768 > $OUT/prof_small.log 2>&1; rc=$?; echo "prof_small rc=$rc" | tee -a $OUT/summary.txt; cat $OUT/prof_small.log | cut -c1-200
[ $rc -ne 0 ] && exit $rc
bash tools/pmc_small.sh r03_j/pmc8192 8192 fast 0 0 > $OUT/pmc8192.log 2>&1; rc=$?; echo "pmc 8192 rc=$rc" | tee -a $OUT/summary.txt
[ $rc -ne 0 ] && exit $rc
bash tools/pmc_small.sh r03_j/pmc16384 16384 fast 0 0 > $OUT/pmc16384.log 2>&1; rc=$?; echo "pmc 16384 rc=$rc" | tee -a $OUT/summary.txt
exit $rc
