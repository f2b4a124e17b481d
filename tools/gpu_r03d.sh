#!/bin/bash
# round-3 visit D: balanced-run kernel (inbox layout) at small/mid N; row-accumulating tasks at large N
set -o pipefail
OUT=gpurun_out/r03_e
mkdir -p $OUT
for n in 8192 16384 6144 12288 24576 32768 49152; do
  timeout -k 10 300 ./build/balbench $n > $OUT/balbench_$n.txt 2>&1; rc=$?
  echo "balbench $n rc=$rc" | tee -a $OUT/summary.txt
  [ $rc -ne 0 ] && { tail -5 $OUT/balbench_$n.txt; exit $rc; }
done
cat $OUT/balbench_8192.txt $OUT/balbench_16384.txt | cut -c1-250




exit $rc
