#!/bin/bash
# round-3 visit A: fake-host RCCL probe, the sharded GPU tests, the default bench
set -o pipefail
OUT=gpurun_out/r03_a
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/rccl_hostid_probe.py > $OUT/hostid_probe.txt 2>&1
rc=$?; echo "hostid probe rc=$rc" | tee -a $OUT/summary.txt; tail -5 $OUT/hostid_probe.txt
[ $rc -ge 124 ] && exit $rc
timeout -k 10 1500 python -m pytest tests/test_gpu_sharded.py tests/test_validation_dropin.py -m gpu -q -x > $OUT/pytest_sharded.txt 2>&1
rc=$?; echo "pytest sharded rc=$rc" | tee -a $OUT/summary.txt; tail -15 $OUT/pytest_sharded.txt
[ $rc -ge 124 ] && exit $rc
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err
rc=$?; echo "bench rc=$rc" | tee -a $OUT/summary.txt; cat $OUT/bench.json | cut -c1-600
exit $rc
