#!/bin/bash
# round-3 visit H: full GPU tests with the balanced runs wired in, the small-N probe over the auto choice, symbench with the square-only kernel
set -o pipefail
OUT=gpurun_out/r03_h
mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.txt 2>&1; rc=$?
echo "pytest rc=$rc" | tee -a $OUT/summary.txt; tail -8 $OUT/pytest_gpu.txt
[ $rc -ne 0 ] && exit $rc
PROBE_SHAPES=1x2,1x4,2x4,1x8,1x10 timeout -k 10 400 python tools/smalln_probe.py 4096 6144 8192 12288 16384 24576 32768 49152 65536 > $OUT/smalln.jsonl 2> $OUT/smalln.err; rc=$?
echo "smalln rc=$rc" | tee -a $OUT/summary.txt
[ $rc -ne 0 ] && { tail -5 $OUT/smalln.err; exit $rc; }
timeout -k 10 500 ./build/symbench 262144 7 > $OUT/symbench_262144.txt 2>&1; rc=$?; echo "symbench rc=$rc" | tee -a $OUT/summary.txt
grep -E "^sym|^one-sided|force alone|R=1" $OUT/symbench_262144.txt | cut -c1-150
exit $rc
