#!/bin/bash
# rocprofv3 kernel stats + un-profiled bench lines for the other BASELINE configs on one GPU
# (N=65536 fp32, N=262144 fp64, N=1048576 fp32). Usage via gpurun: bash tools/prof_configs.sh <tag>
set -o pipefail
TAG=${1:-cfg}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
bench() { local name=$1; shift; timeout -k 10 400 python3 $REPO/bench.py "$@" > $OUT/${name}_bench.json 2> $OUT/${name}_bench.err; echo "$name bench rc=$?"; }
prof() { local name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 $REPO/bench.py --no-cpu-baseline --no-general-path --repeats 2 "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name prof rc=$?"; }
bench n65536 --bodies 65536 --steps 1000 --warmup 50 --no-cpu-baseline && \
bench f64 --dtype f64 --steps 20 --warmup 3 && \
bench n1048576 --bodies 1048576 --steps 5 --warmup 1 --no-cpu-baseline && \
prof n65536 --bodies 65536 --steps 200 --warmup 10 && \
prof f64 --dtype f64 --steps 5 --warmup 1 && \
prof n1048576 --bodies 1048576 --steps 3 --warmup 1
