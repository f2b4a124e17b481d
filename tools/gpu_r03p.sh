#!/bin/bash
# round-3 visit P: GPU tests with the fused step wired in; the small-N probe over the automatic choice; the reference's literal loop
set -o pipefail
OUT=gpurun_out/r03_p
mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.txt 2>&1; rc=$?
echo "pytest rc=$rc" | tee -a $OUT/summary.txt; tail -5 $OUT/pytest_gpu.txt
[ $rc -ne 0 ] && exit $rc
PROBE_SHAPES=1x2,1x4 timeout -k 10 600 python tools/smalln_probe.py 512 1024 1536 2048 3072 4096 5120 6144 7168 8192 8704 9216 10240 12288 16384 24576 32768 > $OUT/smalln.jsonl 2> $OUT/smalln.err; rc=$?
echo "smalln rc=$rc" | tee -a $OUT/summary.txt
[ $rc -ne 0 ] && { tail -5 $OUT/smalln.err; exit $rc; }
./n-bodysimulation_amd/bin/nbody_headless --n 8192 --steps 2000 --init ref --sync-each-step > $OUT/headless_sync_8192.txt 2>&1; echo "headless sync rc=$?" | tee -a $OUT/summary.txt
./n-bodysimulation_amd/bin/nbody_headless --n 8192 --steps 2000 --init ref > $OUT/headless_queued_8192.txt 2>&1; echo "headless queued rc=$?" | tee -a $OUT/summary.txt
tail -1 $OUT/headless_sync_8192.txt | cut -c1-200; tail -1 $OUT/headless_queued_8192.txt | cut -c1-200
bash tools/prof_small.sh r03_p/small 8192 4096 2048 > $OUT/prof_small.log 2>&1; echo "prof_small rc=$?" | tee -a $OUT/summary.txt; cut -c1-200 $OUT/prof_small.log
python - <<PY
import json
for ln in open("$OUT/smalln.jsonl"):
    r=json.loads(ln)
    best=sorted(((v['us_per_step'],k) for k,v in r.items() if isinstance(v,dict)))[:3]
    fa=r['fast_auto']
    print(r['n'], 'fast_auto %.2f us %.3e'%(fa['us_per_step'],fa['pairs_per_s']), {k:fa['info'][k] for k in ('fused','balanced','block_bodies','workgroups')}, '| best:', [(k,u) for u,k in best])
PY
