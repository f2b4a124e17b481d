#!/bin/bash
# Same-box A/B of two builds of tools/symbench.hip (build/symbench_r04: the round-4 device header; build/symbench: this tree), alternately,
# N = 262144: square / general kernels on both pair-arithmetic paths, and the rectangular launch.   usage: gpu_symbench_ab.sh <tag> [passes]
set -o pipefail
TAG=${1:-ab}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for rep in $(seq 1 ${2:-2}); do
  order="symbench_r04 symbench"; [ $((rep % 2)) -eq 0 ] && order="symbench symbench_r04"   # alternate who goes first: clock / thermal drift cancels
  for b in $order; do
    echo "== $b (pass $rep)" >> $OUT/symbench_ab.txt
    SYMBENCH_EQ=1 timeout -k 10 200 build/$b 262144 7 2>&1 | grep -E "SQUARE (general|equal)|general kernel|bpl8" >> $OUT/symbench_ab.txt || exit $?
    echo "== $b rect (pass $rep)" >> $OUT/symbench_ab.txt
    SYMBENCH_RECT=1 timeout -k 10 200 build/$b 262144 7 2>&1 | grep -E "^rect" >> $OUT/symbench_ab.txt || exit $?
  done
done
python3 - $OUT/symbench_ab.txt <<'PY' | tee $OUT/symbench_ab_table.txt
import re, sys, collections
cur = None; t = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    m = re.match(r"== (\S+)", ln)
    if m: cur = m.group(1); continue
    m = re.match(r"(sym .*?)\s+[\d.]+ ms \(force alone ([\d.]+)\)", ln)
    if m: t[(m.group(1).strip(), cur)].append(float(m.group(2))); continue
    m = re.match(r"rect .*?, (\S+) path: general kernel ([\d.]+) ms, rect-only kernel ([\d.]+) ms", ln)
    if m:
        t[("rect launch, general kernel, %s path" % m.group(1), cur)].append(float(m.group(2)))
        t[("rect launch, rect-only kernel, %s path" % m.group(1), cur)].append(float(m.group(3)))
names = sorted({k[0] for k in t})
print("%-50s %12s %12s %8s" % ("force launch alone, ms (mean of all passes)", "round-4 hdr", "this tree", "delta"))
for n in names:
    a, b = t.get((n, "symbench_r04"), []), t.get((n, "symbench"), [])
    if a and b:
        ma, mb = sum(a) / len(a), sum(b) / len(b)
        print("%-50s %12.3f %12.3f %+7.2f%%" % (n, ma, mb, (mb / ma - 1) * 100))
PY
