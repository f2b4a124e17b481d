#!/bin/bash
# Round-5 visit c:
#  (1) ONE RANK of BASELINE configs[3] alone on the GPU, kernel by kernel: tools/rank_probe.py (the sharded step of rank G/2 of G with
#      no-op collectives) under rocprofv3 --kernel-trace --stats, G = 8 and G = 1, general pair arithmetic and equal-mass path;
#  (2) same-box A/B of the one-template rotation body against round 4's three copies: build/symbench_r04 (round-4 header) and
#      build/symbench (this tree) alternately, N = 262144, square / general kernels on both paths, and the rectangular launch.
set -o pipefail
TAG=${1:-r05c}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for mode in general eq; do
  flag=""; [ $mode = general ] && flag="--no-equal-mass"
  for g in 8 1; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rank_${mode}_g$g -- python3 $REPO/tools/rank_probe.py --steps 4 $flag $g > $OUT/rank_${mode}_g$g.json 2> $OUT/rank_${mode}_g$g.err || exit $?
    cat $OUT/rank_${mode}_g$g.json
    find $OUT/rank_${mode}_g$g -name "*kernel_stats.csv" | head -1 | xargs -r head -9
  done
done
cd $REPO
for rep in 1 2; do
  for b in symbench_r04 symbench; do
    echo "== $b (pass $rep)" >> $OUT/symbench_ab.txt
    SYMBENCH_EQ=1 timeout -k 10 200 build/$b 262144 7 2>&1 | grep -E "SQUARE (general|equal)|general kernel|bpl8" >> $OUT/symbench_ab.txt || exit $?
    echo "== $b rect (pass $rep)" >> $OUT/symbench_ab.txt
    SYMBENCH_RECT=1 timeout -k 10 200 build/$b 262144 7 2>&1 | grep -E "^rect" >> $OUT/symbench_ab.txt || exit $?
  done
done
cat $OUT/symbench_ab.txt
