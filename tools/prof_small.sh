#!/bin/bash
# rocprofv3 kernel stats of queued steps at small / mid N with the automatically chosen kernel. Usage via gpurun:
#   bash tools/prof_small.sh <tag> [N ...]
set -o pipefail
TAG=${1:-small}; shift
REPO=$PWD; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cat > $OUT/run_small.py <<PY
import sys
sys.path.insert(0, "$REPO")
import nbody_amd
n = int(sys.argv[1])
sim = nbody_amd.engine.Simulation(nbody_amd.engine.seeded_bodies(n, 1, 1), dt=0.01, eps2=0.002)
print(sim.ctx.step_info(n))
sim.run(300); sim.ctx.sync()
PY
cd /tmp
for N in ${@:-8192 16384 32768 65536}; do
  name=n${N}_auto
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 $OUT/run_small.py $N > $OUT/$name.log 2>&1 || exit 1
  echo "== $name $(grep symmetric $OUT/$name.log | cut -c1-150)"; find $OUT/$name -name "*kernel_stats.csv" | head -1 | xargs -r head -3 | cut -c1-150
done
