#!/bin/bash
# One GPU-box visit, in stages. Usage (from the repo root, through gpurun):
#     bash tools/gpu_round.sh <tag> [stage ...]          default stages: suite smoke bench prof
# Stages are chained in the order given: one that fails or times out stops the visit (no GPU step after a killed one). Everything
# lands under gpurun_out/<tag>/; copy what is to be judged into profiles/.
#   suite     python -m pytest tests -m gpu -q -x
#   smoke     __graft_entry__.smoke()
#   bench     python bench.py --steps 20 --warmup 5                                    (the driver's N = 1 command)
#   prof      rocprofv3 --kernel-trace --stats of the same bench command (3 repeats, no CPU baseline, no extras)
#   configs   bench lines + kernel stats of the other BASELINE configs on one GPU (N = 65536, fp64, N = 1048576)
#   sizes     ONE box, general pair arithmetic, queued steps: bench lines at N = 4096, 8192, 16384, 32768, 45056, 65536, 131072, 160000, 262144,
#             1048576 (+ the default bench with --masses random), then tools/sizes_table.py: fraction of peak, decomposition, shader clock per size
#   pmcn      PMC passes (SQ group, GRBM) of bench.py at the sizes in $PMC_N (default "16384 32768": the two rows furthest below the roofline)
#   sync      tools/sync_probe (N = 8192 and 16384) + nbody_headless --sync-each-step: what a synchronous simulate() per step costs
#   rank      ONE rank of configs[3] alone (tools/rank_probe.py under rocprofv3 --stats; G = 8 and G = 1; general and equal-mass path)
#   local8    nbody_headless --ngpu 8 --transport local --share-devices at N = 1048576 under rocprofv3 --stats (and --ngpu 1)
#   symab     same-box A/B of build/symbench_r04 (an older device header, see below) against build/symbench: NEEDS both binaries
#   pmc       PMC passes (one counter group per run, --kernel-trace only beside --pmc): the bench's general path, its equal-mass path
#             (--equal-mass auto), the same step with the block sums added in place (--inplace-sums on: SQ, GRBM, FETCH_SIZE, WRITE_SIZE),
#             and tools/sync_probe at N = 8192 (the two fused kernels); then tools/pmc_summary.py
#   rehearse  bench.py --gpus 4 --fake-hosts at N = 1048576 (four RCCL ranks on the one GPU), both transports, started WITHOUT a launcher
#   pkbank    tools/pkbank_mb: does the VGPR bank of a packed instruction's operands change its issue cost?
#   multirank only the multi-rank files of the GPU suite (tests/test_gpu_sharded.py, tests/test_zz_rccl_rehearsal.py)
#   contend   three processes stepping the reference's loop while a fourth keeps the GPU full; results compared byte for byte
# build/sync_probe and build/symbench come from `make tools` (or the hipcc lines at the top of those files); an A/B partner for
# `symab` is built from an older header:  git show <rev>:n-bodysimulation_amd/csrc/nbody_kernels.hip.h > /tmp/old/n-bodysimulation_amd/csrc/...
set -o pipefail
TAG=${1:-visit}; shift
STAGES=${@:-suite smoke bench prof}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
HEADLESS=$REPO/n-bodysimulation_amd/bin/nbody_headless
say() { echo "$@" | tee -a $OUT/summary.txt; }
stats_head() { find $1 -name "*kernel_stats.csv" | head -1 | xargs -r head -${2:-8}; }
prof() {  # <dir name> <program and arguments ...>: rocprofv3 kernel stats; the program itself goes after `--`
  local name=$1; shift
  ( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- "$@" > $OUT/$name.out 2> $OUT/$name.err )
}
pmc() {   # <dir name> <counters, comma separated> <program and arguments ...>
  local name=$1 counters=${2//,/ }; shift 2
  ( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $OUT/$name -- "$@" > $OUT/$name.out 2> $OUT/$name.err )
}
SQ1=SQ_WAVES,SQ_INSTS_VALU,SQ_ACTIVE_INST_VALU,SQ_BUSY_CYCLES,SQ_WAVE_CYCLES,SQ_INSTS_LDS,SQ_INSTS_SALU,SQ_WAIT_INST_ANY
SQ2=SQ_ACTIVE_INST_ANY,SQ_WAIT_ANY,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_INST_CYCLES_VMEM,SQ_ACTIVE_INST_LDS,SQ_WAIT_INST_LDS
GRBM=GRBM_GUI_ACTIVE,GRBM_COUNT

for stage in $STAGES; do
  rc=0
  case $stage in
  suite)
    timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.txt 2>&1; rc=$?; tail -5 $OUT/pytest_gpu.txt ;;
  smoke)
    timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; rc=$?; tail -2 $OUT/smoke.txt ;;
  bench)
    timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; rc=$?; cat $OUT/bench.json ;;
  prof)
    prof prof python3 $REPO/bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-equal-mass-extras; rc=$?
    cp $OUT/prof.out $OUT/bench_prof.json; stats_head $OUT/prof ;;
  configs)
    timeout -k 10 400 python3 bench.py --bodies 65536 --steps 1000 --warmup 50 --no-cpu-baseline > $OUT/n65536_bench.json 2> $OUT/n65536_bench.err && \
    timeout -k 10 400 python3 bench.py --dtype f64 --steps 20 --warmup 3 > $OUT/f64_bench.json 2> $OUT/f64_bench.err && \
    timeout -k 10 400 python3 bench.py --bodies 1048576 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/n1048576_bench.json 2> $OUT/n1048576_bench.err && \
    prof prof_n65536 python3 $REPO/bench.py --no-cpu-baseline --no-equal-mass-extras --repeats 2 --bodies 65536 --steps 200 --warmup 10 && \
    prof prof_f64 python3 $REPO/bench.py --no-cpu-baseline --no-equal-mass-extras --repeats 2 --dtype f64 --steps 5 --warmup 1 && \
    prof prof_n1048576 python3 $REPO/bench.py --no-cpu-baseline --no-equal-mass-extras --repeats 2 --bodies 1048576 --steps 3 --warmup 1; rc=$?
    for f in n65536 f64 n1048576; do stats_head $OUT/prof_$f 3; done ;;
  sizes)
    for spec in 4096:20000:500 8192:10000:500 16384:4000:200 32768:1500:100 45056:1000:50 65536:400:40 131072:100:10 160000:60:6 262144:20:5 1048576:3:1; do
      IFS=: read n k w <<< "$spec"
      [ $rc -eq 0 ] || break
      timeout -k 10 300 python3 bench.py --bodies $n --steps $k --warmup $w --min-seconds 4 --no-cpu-baseline > $OUT/n${n}_bench.json 2> $OUT/n${n}_bench.err; rc=$?
    done
    for spec in 65536:400:40 131072:100:10 160000:60:6 262144:20:5 1048576:3:1; do   # the same sizes with the block sums added in place (no slabs)
      IFS=: read n k w <<< "$spec"
      [ $rc -eq 0 ] || break
      timeout -k 10 300 python3 bench.py --bodies $n --steps $k --warmup $w --min-seconds 4 --no-cpu-baseline --inplace-sums on > $OUT/n${n}_inplace_bench.json 2> $OUT/n${n}_inplace_bench.err; rc=$?
    done
    [ $rc -eq 0 ] && { timeout -k 10 300 python3 bench.py --masses random --no-cpu-baseline > $OUT/n262144_random_masses_bench.json 2> $OUT/n262144_random_masses_bench.err; rc=$?; }
    python3 tools/sizes_table.py $OUT | tee $OUT/sizes_table.md ;;
  pmcn)
    for n in ${PMC_N:-16384 32768}; do
      [ $rc -eq 0 ] || break
      k=$((400000000 / n / (n / 4096))); [ $k -gt 2000 ] && k=2000; [ $k -lt 3 ] && k=3
      B="python3 $REPO/bench.py --bodies $n --steps $k --warmup 2 --repeats 1 --no-cpu-baseline --no-equal-mass-extras --no-clock"
      pmc n${n}_sq1 $SQ1 $B && pmc n${n}_grbm $GRBM $B; rc=$?
    done
    python3 tools/pmc_summary.py $OUT $OUT/summary ;;
  sync)
    for n in 8192 16384; do timeout -k 10 120 build/sync_probe $n 3000 > $OUT/sync_probe_n$n.txt 2>&1 || rc=$?; done
    grep -E "simulate\(\) per step|queued: nbody_step" $OUT/sync_probe_n8192.txt $OUT/sync_probe_n16384.txt
    for k in 1 2 3; do [ $rc -eq 0 ] && { timeout -k 10 120 $HEADLESS --n 8192 --steps 20000 --init libc --sync-each-step | tail -1 >> $OUT/headless_sync_each_step_n8192.txt || rc=$?; }; done
    cat $OUT/headless_sync_each_step_n8192.txt ;;
  rank)
    for mode in general eq; do
      flag=""; [ $mode = general ] && flag="--no-equal-mass"
      for g in 8 1; do
        [ $rc -eq 0 ] || break
        prof rank_${mode}_g$g python3 $REPO/tools/rank_probe.py --steps 4 $flag $g; rc=$?
        cat $OUT/rank_${mode}_g$g.out; stats_head $OUT/rank_${mode}_g$g 9
      done
    done ;;
  local8)
    for mode in general eq; do
      flag=""; [ $mode = general ] && flag="--no-equal-mass"
      for g in 8 1; do
        [ $rc -eq 0 ] || break
        extra=""; [ $g -gt 1 ] && extra="--transport local --share-devices"
        prof local_${mode}_g$g $HEADLESS --n 1048576 --steps 4 --dt 0.01 --init plummer $flag --ngpu $g $extra; rc=$?
        tail -1 $OUT/local_${mode}_g$g.out; stats_head $OUT/local_${mode}_g$g 10
      done
    done ;;
  symab)
    for rep in 1 2 3 4; do
      order="symbench_r04 symbench"; [ $((rep % 2)) -eq 0 ] && order="symbench symbench_r04"   # alternate who goes first: clock / thermal drift cancels
      for b in $order; do
        [ $rc -eq 0 ] || break
        echo "== $b (pass $rep)" >> $OUT/symbench_ab.txt
        SYMBENCH_EQ=1 timeout -k 10 200 build/$b 262144 7 2>&1 | grep -E "SQUARE (general|equal)|general kernel|bpl8" >> $OUT/symbench_ab.txt || rc=$?
        echo "== $b rect (pass $rep)" >> $OUT/symbench_ab.txt
        SYMBENCH_RECT=1 timeout -k 10 200 build/$b 262144 7 2>&1 | grep -E "^rect" >> $OUT/symbench_ab.txt || rc=$?
      done
    done
    python3 tools/symbench_ab_table.py $OUT/symbench_ab.txt | tee $OUT/symbench_ab_table.txt ;;
  pmc)
    B="python3 $REPO/bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-equal-mass-extras --no-clock"
    pmc gen_sq1 $SQ1 $B && pmc gen_grbm $GRBM $B && pmc gen_fetch FETCH_SIZE $B && pmc gen_write WRITE_SIZE $B && \
    pmc eq_sq1 $SQ1 $B --equal-mass auto && pmc eq_grbm $GRBM $B --equal-mass auto && \
    pmc inplace_sq1 $SQ1 $B --inplace-sums on && pmc inplace_grbm $GRBM $B --inplace-sums on && \
    pmc inplace_fetch FETCH_SIZE $B --inplace-sums on && pmc inplace_write WRITE_SIZE $B --inplace-sums on && \
    pmc fused_sq1 $SQ1 $REPO/build/sync_probe 8192 300 && pmc fused_grbm $GRBM $REPO/build/sync_probe 8192 300 && pmc fused_sq2 $SQ2 $REPO/build/sync_probe 8192 300; rc=$?
    python3 tools/pmc_summary.py $OUT $OUT/summary ;;
  rehearse)
    for comm in torch native; do
      [ $rc -eq 0 ] || break
      timeout -k 10 500 python bench.py --gpus 4 --fake-hosts --comm $comm --steps 3 --warmup 1 --repeats 3 > $OUT/rehearsal_4ranks_n1048576_$comm.json 2> $OUT/rehearsal_4ranks_n1048576_$comm.err; rc=$?
    done ;;
  contend)
    base="--n 8192 --steps 1500 --init libc --sync-each-step"
    $HEADLESS $base --dump $OUT/alone | tail -1 > $OUT/alone.json
    for r in 1 2 3 4 5; do
      $HEADLESS --n 131072 --steps 1200 --init plummer --dt 0.01 --quiet > /dev/null 2>&1 &
      HOG=$!
      for k in 1 2 3; do $HEADLESS $base --dump $OUT/s${k} | tail -1 > $OUT/s${k}.json & P[$k]=$!; done
      for k in 1 2 3; do wait ${P[$k]}; done
      wait $HOG
      for k in 1 2 3; do
        same=yes; for ext in x v a; do cmp -s $OUT/alone.$ext.f4 $OUT/s${k}.$ext.f4 || same=NO; done
        say "contention round $r proc $k identical=$same $(python3 -c "import json;d=json.load(open('$OUT/s${k}.json'));print('fallback_waves',d['inplace_fallback_waves'],'us/step %.0f'%(d['seconds']/d['steps']*1e6))")"
      done
    done
    find "$OUT" -maxdepth 1 -name "*.f4" -delete
    bad=$(grep -c "identical=NO" $OUT/summary.txt); say "mismatches: $bad"; [ "$bad" = 0 ] || rc=1 ;;
  pkbank)
    timeout -k 10 120 build/pkbank_mb > $OUT/pkbank_mb.txt 2>&1; rc=$?; cat $OUT/pkbank_mb.txt ;;
  multirank)
    timeout -k 10 900 python -m pytest tests/test_gpu_sharded.py tests/test_zz_rccl_rehearsal.py -q -x > $OUT/pytest_multirank.txt 2>&1; rc=$?; tail -5 $OUT/pytest_multirank.txt ;;
  *) say "unknown stage $stage"; exit 2 ;;
  esac
  say "$stage rc=$rc"
  [ $rc -eq 0 ] || exit $rc
done
exit 0
