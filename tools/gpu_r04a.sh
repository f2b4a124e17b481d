#!/bin/bash
# Round-4 visit A: the GPU suite with the rehearsals last, then the DIAGNOSTIC of the multi-rank one-GPU rehearsal
# (VERDICT r03 weak 1/2): the same bench.py run with 3 and 4 ranks under different numbers of hardware queues per process
# (communication stream priority, GPU_MAX_HW_QUEUES, a bystander process that holds queues like the test runner does).
set -o pipefail
OUT=gpurun_out/r04a
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.txt 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $OUT/summary.txt; tail -5 $OUT/pytest_gpu.txt
probe() {   # tag, ranks, bodies, steps, extra env..., then bench args after --
  tag=$1; ranks=$2; bodies=$3; steps=$4; shift 4
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  port=$((29600 + RANDOM % 300))
  env "${envs[@]}" NBODY_BENCH_STACKS_AFTER=150 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $ranks --master-addr 127.0.0.1 --master-port $port \
      bench.py --gpus $ranks --fake-hosts --comm native --bodies $bodies --steps $steps --warmup 2 --repeats 3 --no-general-path "$@" > $OUT/$tag.json 2> $OUT/$tag.err
  rc=$?
  echo "$tag rc=$rc" | tee -a $OUT/summary.txt
  python - "$OUT/$tag.json" "$tag" <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    c = d["config"]
    print(sys.argv[2], "ms/step %.2f (min %.2f max %.2f)" % (d["ms_per_step"], d["ms_per_step_min"], d["ms_per_step_max"]),
          "kernel_ms %.2f" % d["roofline"]["kernel_ms_per_step"],
          "gather %.2f exposed %.2f exch %.2f exposed %.2f" % tuple(c["comm_rank0"][k] for k in ("all_gather_ms_avg", "exposed_ms_avg", "exchange_ms_avg", "exchange_exposed_ms_avg")),
          "prio", c["rccl"].get("comm_priority"), "queues", [(q.get("total"), {k: v for k, v in q.items() if k.startswith("type_")}) for q in c["rccl"].get("kfd_queues", [])],
          "single", d.get("single_gpu_same_n", {}).get("ms_per_step"))
except Exception as e:
    print(sys.argv[2], "no line:", repr(e))
PY
  return $rc
}
probe r3_high        3 49152 10 A=1 -- --comm-priority high   || exit 1
probe r3_normal      3 49152 10 A=1 -- --comm-priority normal || exit 1
probe r3_normal_hwq2 3 49152 10 GPU_MAX_HW_QUEUES=2 -- --comm-priority normal || exit 1
python tools/queue_holder.py 8 200 > $OUT/holder.txt 2>&1 &
HOLDER=$!
sleep 20
probe r3_high_holder   3 49152 10 A=1 -- --comm-priority high   || { kill $HOLDER; exit 1; }
probe r3_normal_holder 3 49152 10 A=1 -- --comm-priority normal || { kill $HOLDER; exit 1; }
kill $HOLDER; wait $HOLDER 2>/dev/null
cat $OUT/holder.txt | tee -a $OUT/summary.txt
probe r4_high        4 262144 5 A=1 -- --comm-priority high   || exit 1
probe r4_normal      4 262144 5 A=1 -- --comm-priority normal || exit 1
probe r4_normal_hwq2 4 262144 5 GPU_MAX_HW_QUEUES=2 -- --comm-priority normal || exit 1
probe r2_normal      2 262144 5 A=1 -- --comm-priority normal || exit 1
ls /sys/class/kfd/kfd/proc 2>&1 | head -5 >> $OUT/summary.txt
cat /sys/module/amdgpu/parameters/hws_max_conc_proc /sys/module/amdgpu/parameters/sched_policy /sys/module/amdgpu/parameters/num_kcq 2>&1 | tr '\n' ' ' >> $OUT/summary.txt
echo >> $OUT/summary.txt
exit 0
