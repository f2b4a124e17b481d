#!/bin/bash
# rocprofv3 kernel stats for the other BASELINE configs (N=65536 fp32, N=262144 fp64, N=1048576 fp32 on one GPU).
set -o pipefail
TAG=${1:-cfg}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() { local name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 $REPO/bench.py --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; }
run n65536 --bodies 65536 --steps 200 --warmup 10 && \
run f64 --dtype f64 --steps 5 --warmup 1 && \
run n1048576 --bodies 1048576 --steps 3 --warmup 1
