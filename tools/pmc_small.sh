#!/bin/bash
# PMC passes (SQ + GRBM) over queued steps at small N. Usage via gpurun: bash tools/pmc_small.sh <tag> N kernel waves bpl
set -o pipefail
TAG=$1; N=$2; K=$3; W=$4; B=$5
REPO=$PWD; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cat > $OUT/run_small.py <<PY
import sys
sys.path.insert(0, "$REPO")
import nbody_amd
n, kernel, w, b = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
k = {"onesided": nbody_amd.KERNEL_ONESIDED, "symmetric": nbody_amd.KERNEL_SYMMETRIC, "fast": nbody_amd.KERNEL_FAST}[kernel]
sim = nbody_amd.engine.Simulation(nbody_amd.engine.seeded_bodies(n, 1, 1), dt=0.01, eps2=0.002, kernel=k)
if w: sim.ctx.set_symmetric_shape(w, b); sim.ctx.reserve(n)
sim.run(40); sim.ctx.sync()
PY
cd /tmp
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $OUT/run_small.py $N $K $W $B > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY && \
run sq2 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_IFETCH SQ_WAIT_INST_LDS && \
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
