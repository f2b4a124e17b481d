#!/bin/bash
# round-3 visit C: balanced-run kernel bench at small/mid N, then the full GPU test suite
set -o pipefail
OUT=gpurun_out/r03_c
mkdir -p $OUT
for n in 8192 16384 4096 32768 65536; do
  timeout -k 10 300 ./build/balbench $n > $OUT/balbench_$n.txt 2>&1; rc=$?
  echo "balbench $n rc=$rc" | tee -a $OUT/summary.txt
  [ $rc -ne 0 ] && { tail -5 $OUT/balbench_$n.txt; exit $rc; }
done
cat $OUT/balbench_8192.txt $OUT/balbench_16384.txt | cut -c1-260
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.txt 2>&1; rc=$?
echo "pytest rc=$rc" | tee -a $OUT/summary.txt; tail -8 $OUT/pytest_gpu.txt
exit $rc
