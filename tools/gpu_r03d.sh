#!/bin/bash
# round-3 visit D: balanced-run kernel (inbox layout) at small/mid N; row-accumulating tasks at large N
set -o pipefail
OUT=gpurun_out/r03_d
mkdir -p $OUT
for n in 8192 16384 4096 6144 12288 24576 32768; do
  timeout -k 10 300 ./build/balbench $n > $OUT/balbench_$n.txt 2>&1; rc=$?
  echo "balbench $n rc=$rc" | tee -a $OUT/summary.txt
  [ $rc -ne 0 ] && { tail -5 $OUT/balbench_$n.txt; exit $rc; }
done
cat $OUT/balbench_8192.txt $OUT/balbench_16384.txt | cut -c1-250
timeout -k 10 400 ./build/symbench 262144 5 > $OUT/symbench_rows_262144.txt 2>&1; rc=$?; echo "symbench 262144 rc=$rc" | tee -a $OUT/summary.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 ./build/symbench 1048576 3 > $OUT/symbench_rows_1048576.txt 2>&1; rc=$?; echo "symbench 1048576 rc=$rc" | tee -a $OUT/summary.txt
grep -A8 "row-accumulating" $OUT/symbench_rows_*.txt | cut -c1-300
exit $rc
