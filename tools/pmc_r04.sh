#!/bin/bash
# Round-4 PMC passes (one counter group per run; --kernel-trace only beside --pmc): the default bench (equal-mass path), the same with
# --masses random (general path), and tools/sync_probe at N = 8192 (the two-array and the in-place fused step side by side).
set -o pipefail
TAG=${1:-r04_pmc}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
bench() {  # name, bench args (quoted), counters...
  local name=$1 bargs=$2; shift 2
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $REPO/bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-general-path $bargs > $OUT/$name.json 2> $OUT/$name.err
  echo "$name rc=$?"
}
probe() {  # name, counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- $REPO/build/sync_probe 8192 300 > $OUT/$name.txt 2> $OUT/$name.err
  echo "$name rc=$?"
}
bench eq_sq1 "" SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY && \
bench eq_grbm "" GRBM_GUI_ACTIVE GRBM_COUNT && \
bench eq_fetch "" FETCH_SIZE && \
bench eq_write "" WRITE_SIZE && \
bench gen_sq1 "--masses random" SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY && \
bench gen_grbm "--masses random" GRBM_GUI_ACTIVE GRBM_COUNT && \
probe fused_sq1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY && \
probe fused_grbm GRBM_GUI_ACTIVE GRBM_COUNT && \
probe fused_sq2 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
cd $REPO && python3 tools/pmc_summary.py gpurun_out/$TAG gpurun_out/$TAG/summary
ls $OUT | head -40
