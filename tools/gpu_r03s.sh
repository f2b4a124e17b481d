#!/bin/bash
# Visit after the equal-mass path: smoke, bench lines (default / path off / random masses / other sizes), rocprof stats, 2-rank rehearsal.
set -o pipefail
OUT=gpurun_out/r03s
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
step() { echo "== $1"; }
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; rc=$?; echo "smoke rc=$rc"; tail -2 $OUT/smoke.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; rc=$?; echo "bench rc=$rc"; [ $rc -eq 0 ] || { tail -5 $OUT/bench_default.err; exit $rc; }
timeout -k 10 300 python bench.py --no-cpu-baseline --equal-mass off > $OUT/bench_eq_off.json 2> $OUT/bench_eq_off.err; rc=$?; echo "bench eq off rc=$rc"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline --masses random > $OUT/bench_random_masses.json 2> $OUT/bench_random.err; rc=$?; echo "bench random rc=$rc"; [ $rc -eq 0 ] || exit $rc
for n in 16384 32768 65536 1048576; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --bodies $n --min-seconds 4 > $OUT/bench_n$n.json 2> $OUT/bench_n$n.err; rc=$?; echo "bench $n rc=$rc"; [ $rc -eq 0 ] || exit $rc
done
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -- python3 $REPO/bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline > $REPO/$OUT/bench_prof.json 2> $REPO/$OUT/prof.err ); rc=$?; echo "rocprof rc=$rc"; [ $rc -eq 0 ] || exit $rc
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -6
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29633 bench.py --gpus 2 --fake-hosts --bodies 262144 --steps 5 --warmup 2 --repeats 3 > $OUT/bench_2ranks.json 2> $OUT/bench_2ranks.err; rc=$?; echo "2 ranks rc=$rc"; [ $rc -eq 0 ] || { tail -5 $OUT/bench_2ranks.err; exit $rc; }
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03s/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unparsed", e); continue
    em=d["config"].get("equal_mass",{})
    print(f.split("/")[-1], "N",d["config"]["n_bodies"], "ms/step %.4f"%d["ms_per_step"], "value %.3e"%d["value"], "frac %.3f"%d["roofline"]["frac"], "eq",em.get("path_taken"), "general", (em.get("general_path") or {}).get("ms_per_step"))
PY
