"""Mean force-launch time per variant and build from the lines tools/gpu_round.sh's `symab` stage collects (two builds of
tools/symbench.hip run alternately on one box).  usage: symbench_ab_table.py gpurun_out/<tag>/symbench_ab.txt"""
import collections
import re
import sys

cur = None
t = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    m = re.match(r"== (\S+)", ln)
    if m:
        cur = m.group(1)
        continue
    m = re.match(r"(sym .*?)\s+[\d.]+ ms \(force alone ([\d.]+)\)", ln)
    if m:
        t[(m.group(1).strip(), cur)].append(float(m.group(2)))
        continue
    m = re.match(r"rect .*?, (\S+) path: general kernel ([\d.]+) ms, rect-only kernel ([\d.]+) ms", ln)
    if m:
        t[("rect launch, general kernel, %s path" % m.group(1), cur)].append(float(m.group(2)))
        t[("rect launch, rect-only kernel, %s path" % m.group(1), cur)].append(float(m.group(3)))
print("%-50s %12s %12s %8s" % ("force launch alone, ms (mean of all passes)", "symbench_r04", "symbench", "delta"))
for n in sorted({k[0] for k in t}):
    a, b = t.get((n, "symbench_r04"), []), t.get((n, "symbench"), [])
    if a and b:
        ma, mb = sum(a) / len(a), sum(b) / len(b)
        print("%-50s %12.3f %12.3f %+7.2f%%" % (n, ma, mb, (mb / ma - 1) * 100))
