#!/bin/bash
# Is the in-place fused step exact when the GPU is shared (tests/test_gpu_driver.py::test_in_place_step_with_the_gpu_shared_between_processes)?
# Rounds of: one hog (queued steps of 131072 bodies) + three processes stepping the reference's loop at N = 8192, compared byte for byte
# with a run alone — with simulate()'s one-off measurement of the decompositions OFF (the protocol alone) and ON (what it decides under
# contention is printed).   bash tools/contention_probe.sh [rounds]  -> gpurun_out/contention/summary.txt
R=${1:-5}
OUT=gpurun_out/contention
mkdir -p $OUT
D=n-bodysimulation_amd/bin/nbody_headless
base="--n 8192 --steps 1500 --init libc --sync-each-step"
for mode in off on; do
  if [ $mode = off ]; then export NBODY_NO_AUTOTUNE=1; else unset NBODY_NO_AUTOTUNE; fi
  $D $base --dump $OUT/alone_$mode | tail -1 > $OUT/alone_$mode.json
  for r in $(seq 1 $R); do
    $D --n 131072 --steps 1200 --init plummer --dt 0.01 --quiet > /dev/null 2>&1 &
    HOG=$!
    for k in 1 2 3; do $D $base --dump $OUT/s${k} | tail -1 > $OUT/s${k}.json & P[$k]=$!; done
    for k in 1 2 3; do wait ${P[$k]}; done
    wait $HOG
    for k in 1 2 3; do
      same=yes; for ext in x v a; do cmp -s $OUT/alone_$mode.$ext.f4 $OUT/s${k}.$ext.f4 || same=NO; done
      echo "autotune $mode round $r proc $k identical=$same $(python3 -c "import json;d=json.load(open('$OUT/s${k}.json'));print('choice',d['autotuned_choice'],'fallback_waves',d['inplace_fallback_waves'],'us/step %.0f'%(d['seconds']/d['steps']*1e6))")" | tee -a $OUT/summary.txt
    done
  done
done
grep -c "identical=NO" $OUT/summary.txt | sed 's/^/mismatches: /' | tee -a $OUT/summary.txt
rm -f $OUT/*.f4
