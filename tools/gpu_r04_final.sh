#!/bin/bash
# Round-4 evidence visit: whole GPU suite, smoke, the default bench line + rocprofv3 kernel stats of the same command, the general
# path as headline (--masses random), the other BASELINE configs, the reference's literal loop, and 4-rank rehearsals of the
# driver's multi-GPU configuration (N = 1048576) through both transports. Steps are chained: a failing step stops the visit.
set -o pipefail
TAG=${1:-r04_final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
step() { name=$1; shift; "$@"; rc=$?; echo "$name rc=$rc" | tee -a $OUT/summary.txt; return $rc; }
step pytest bash -c "timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.txt 2>&1; rc=\$?; tail -4 $OUT/pytest_gpu.txt; exit \$rc" || exit 1
step smoke bash -c "timeout -k 10 200 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.txt 2>&1; rc=\$?; tail -1 $OUT/smoke.txt; exit \$rc" || exit 1
step bench bash -c "timeout -k 10 400 python bench.py > $OUT/bench_n262144.json 2> $OUT/bench.err" || exit 1
( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -- python3 $REPO/bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-general-path > $REPO/$OUT/bench_prof.json 2> $REPO/$OUT/prof.err ); echo "rocprof rc=$?" | tee -a $OUT/summary.txt
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $OUT/bench_n262144_kernel_stats.csv
step bench_random_masses bash -c "timeout -k 10 400 python bench.py --masses random --no-cpu-baseline > $OUT/bench_n262144_random_masses.json 2>> $OUT/bench.err" || exit 1
step bench_n65536 bash -c "timeout -k 10 300 python bench.py --bodies 65536 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_n65536.json 2>> $OUT/bench.err" || exit 1
step bench_n8192 bash -c "timeout -k 10 300 python bench.py --bodies 8192 --steps 2000 --warmup 100 --no-cpu-baseline > $OUT/bench_n8192.json 2>> $OUT/bench.err" || exit 1
step bench_f64 bash -c "timeout -k 10 300 python bench.py --dtype f64 --steps 10 > $OUT/bench_f64.json 2>> $OUT/bench.err" || exit 1
step bench_n1048576 bash -c "timeout -k 10 300 python bench.py --bodies 1048576 --steps 5 --warmup 2 --repeats 3 --no-cpu-baseline > $OUT/bench_n1048576.json 2>> $OUT/bench.err" || exit 1
for i in 1 2 3; do n-bodysimulation_amd/bin/nbody_headless --n 8192 --steps 20000 --sync-each-step | tail -1 >> $OUT/headless_sync_each_step_n8192.txt; done
n-bodysimulation_amd/bin/nbody_headless --n 8192 --steps 20000 | tail -1 >> $OUT/headless_queued_n8192.txt
rehearse() {   # tag, comm
  port=$((29700 + RANDOM % 200))
  NBODY_BENCH_STACKS_AFTER=250 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port $port \
      bench.py --gpus 4 --fake-hosts --comm $2 --steps 3 --warmup 2 --repeats 3 > $OUT/$1.json 2> $OUT/$1.err
}
step rehearsal_4ranks_n1048576_torch rehearse rehearsal_4ranks_n1048576_torch torch || exit 1
step rehearsal_4ranks_n1048576_native rehearse rehearsal_4ranks_n1048576_native native || exit 1
python - <<PY | tee -a $OUT/summary.txt
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        r = d["roofline"]
        print(f.split("/")[-1], "ms/step %.4f value %.4g frac %.3f general %s eq_path %s" % (d["ms_per_step"], d["value"], r["frac"], r.get("frac_general_path"), d.get("equal_mass_path")),
              ("single_gpu_same_n %.2f ms, check %s / random %s, comm %s" % (d["single_gpu_same_n"]["ms_per_step"], d["config"]["multi_gpu_check"]["max_rel_da"], d["config"]["multi_gpu_check"]["random_masses"]["max_rel_da"], d["config"]["comm_rank0"])) if "single_gpu_same_n" in d else "")
    except Exception as e:
        print(f, "no line", repr(e))
PY
exit 0
