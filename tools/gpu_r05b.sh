#!/bin/bash
# Round-5 visit b: what ONE RANK of BASELINE configs[3] (N = 1048576 over 8 ranks) runs, kernel by kernel — the 8 ranks are threads of
# one process on the one GPU (--transport local --share-devices: peer copies between rank threads instead of RCCL), so rocprofv3 sees
# every rank's launches in one trace — and the single-GPU run of the same system beside it. The program itself goes after `--`.
set -o pipefail
TAG=${1:-r05b}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BIN=$PWD/n-bodysimulation_amd/bin/nbody_headless
COMMON="--n 1048576 --steps 4 --dt 0.01 --init plummer"
cd /tmp
for g in 8 1; do
  extra=""; [ $g -gt 1 ] && extra="--transport local --share-devices"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_g$g -- $BIN $COMMON --ngpu $g $extra > $OUT/headless_g$g.txt 2> $OUT/headless_g$g.err || exit $?
  tail -1 $OUT/headless_g$g.txt
  find $OUT/prof_g$g -name "*kernel_stats.csv" | head -1 | xargs -r head -12
done
# the same with the general pair arithmetic (what unequal masses get: the bench's headline path)
for g in 8 1; do
  extra=""; [ $g -gt 1 ] && extra="--transport local --share-devices"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_general_g$g -- $BIN $COMMON --no-equal-mass --ngpu $g $extra > $OUT/headless_general_g$g.txt 2> $OUT/headless_general_g$g.err || exit $?
  tail -1 $OUT/headless_general_g$g.txt
  find $OUT/prof_general_g$g -name "*kernel_stats.csv" | head -1 | xargs -r head -12
done
