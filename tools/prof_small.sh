#!/bin/bash
# rocprofv3 kernel stats of queued steps at small N (one-sided and symmetric kernels). Usage via gpurun: bash tools/prof_small.sh <tag> [N ...]
set -o pipefail
TAG=${1:-small}; shift
REPO=$PWD; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cat > $OUT/run_small.py <<PY
import sys, os
sys.path.insert(0, "$REPO")
import nbody_amd
n, kernel, w, b = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
k = {"onesided": nbody_amd.KERNEL_ONESIDED, "symmetric": nbody_amd.KERNEL_SYMMETRIC, "fast": nbody_amd.KERNEL_FAST}[kernel]
sim = nbody_amd.engine.Simulation(nbody_amd.engine.seeded_bodies(n, 1, 1), dt=0.01, eps2=0.002, kernel=k)
if w: sim.ctx.set_symmetric_shape(w, b); sim.ctx.reserve(n)
sim.run(200); sim.ctx.sync()
PY
cd /tmp
for N in ${@:-8192 16384}; do
  for CFG in "onesided 0 0" "symmetric 1 2" "symmetric 1 4"; do
    set -- $CFG
    name=n${N}_$1_$2_$3
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 $OUT/run_small.py $N $1 $2 $3 > $OUT/$name.log 2>&1 || exit 1
    echo "== $name"; find $OUT/$name -name "*kernel_stats.csv" | head -1 | xargs -r head -4 | cut -c1-160
  done
done
