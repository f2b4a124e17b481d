#!/bin/bash
# Round-5 first visit: tools/gpu_round.sh (suite, smoke, bench, rocprofv3 stats), then what the drain in the in-place fused step and the
# pointer check in front of the host word cost: tools/sync_probe at N = 8192 and 16384, the headless driver's --sync-each-step.
set -o pipefail
TAG=${1:-r05a}
OUT=gpurun_out/$TAG
bash tools/gpu_round.sh $TAG || exit $?
for n in 8192 16384; do
  timeout -k 10 120 build/sync_probe $n 3000 > $OUT/sync_probe_n$n.txt 2>&1 || exit $?
done
grep -E "simulate\(\) per step|queued: nbody_step" $OUT/sync_probe_n8192.txt $OUT/sync_probe_n16384.txt
for k in 1 2 3; do
  timeout -k 10 120 n-bodysimulation_amd/bin/nbody_headless --n 8192 --steps 20000 --init libc --sync-each-step | tail -1 >> $OUT/headless_sync_each_step_n8192.txt || exit $?
done
cat $OUT/headless_sync_each_step_n8192.txt
