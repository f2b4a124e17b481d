#!/bin/bash
# PMC passes over the bench workload (separate runs per counter group, no trace domains but
# --kernel-trace). Usage through gpurun: bash tools/pmc.sh <tag>
set -o pipefail
TAG=${1:-pmc}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
run() {  # name, counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $REPO/bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-general-path $PMC_BENCH_ARGS > $OUT/$name.json 2> $OUT/$name.err
  echo "$name rc=$?"
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY && \
run sq2 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS && \
run grbm GRBM_GUI_ACTIVE GRBM_COUNT && \
run fetch FETCH_SIZE && \
run write WRITE_SIZE && \
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
ls $OUT
