#!/bin/bash
# round-3 visit M: bench.py with several REAL RCCL ranks on the one GPU (distinct NCCL_HOSTIDs): evidence lines for both transports
set -o pipefail
OUT=gpurun_out/r03_m
mkdir -p $OUT
run() { name=$1; np=$2; shift 2; timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $np --master-addr 127.0.0.1 --master-port $((29700 + np)) bench.py --gpus $np --fake-hosts "$@" > $OUT/$name.json 2> $OUT/$name.err; rc=$?; echo "$name rc=$rc" | tee -a $OUT/summary.txt; return $rc; }
run bench_2ranks_native 2 --comm native --bodies 262144 --steps 5 --warmup 2 --repeats 3 || { tail -5 $OUT/bench_2ranks_native.err; exit 1; }
run bench_2ranks_torch 2 --comm torch --bodies 262144 --steps 5 --warmup 2 --repeats 3 || { tail -5 $OUT/bench_2ranks_torch.err; exit 1; }
run bench_4ranks_native 4 --comm native --bodies 262144 --steps 5 --warmup 2 --repeats 3 || { tail -5 $OUT/bench_4ranks_native.err; exit 1; }
run bench_6ranks_torch 6 --comm torch --bodies 196608 --steps 3 --warmup 2 --repeats 2 || { tail -5 $OUT/bench_6ranks_torch.err; exit 1; }
for f in bench_2ranks_native bench_2ranks_torch bench_4ranks_native bench_6ranks_torch; do python - <<PY
import json
d=json.loads([l for l in open("$OUT/$f.json") if l.startswith("{")][-1])
c=d["config"]
print("$f", "ms/step %.2f"%d["ms_per_step"], c["multi_gpu_check"], {k:round(v,3) if isinstance(v,float) else v for k,v in c["comm_rank0"].items()}, c["rccl"]["world"], c["rccl"]["distinct_devices"], c["launch"]["schedule"])
PY
done
