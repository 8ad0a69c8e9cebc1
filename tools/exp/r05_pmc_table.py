"""Per-kernel table of whatever counters the passes of tools/exp/r05_call1.sh collected (mean per launch over the last 2/3 of a
kernel's launches, summed over the counter's instances as rocprofv3 reports it).  usage: r05_pmc_table.py <dir with pmc_*/>"""
import collections
import csv
import glob
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        per[(r["Kernel_Name"][:64], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, _), c in per.items():
        for n, v in c.items():
            acc[k][n].append(v)
names = sorted({n for c in acc.values() for n in c})
print("kernel".ljust(64), " ".join(n[:26].rjust(26) for n in names))
for k, c in sorted(acc.items()):
    if max(len(v) for v in c.values()) < 20:
        continue
    row = []
    for n in names:
        v = c.get(n, [])
        v = v[len(v) // 3:]
        row.append(("%.4g" % (sum(v) / len(v))) if v else "-")
    print(k.ljust(64), " ".join(x.rjust(26) for x in row))
