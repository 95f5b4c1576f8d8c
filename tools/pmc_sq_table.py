"""Per-kernel averages of the counter_collection.csv files written by tools/pmc_sq.sh -> markdown table on stdout."""
import csv
import glob
import sys
from collections import OrderedDict, defaultdict

root = sys.argv[1]
vals = OrderedDict()
for f in sorted(glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if "at::" in name or "elementwise" in name or "distribution" in name:
            continue
        vals.setdefault(name, defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
ctrs = []
for d in vals.values():
    for c in d:
        if c not in ctrs:
            ctrs.append(c)
print("| kernel | launches | " + " | ".join(ctrs) + " |")
print("|---|---|" + "---|" * len(ctrs))
for k, d in vals.items():
    n = max(len(v) for v in d.values())
    print("| `%s` | %d | " % (k[:90], n) + " | ".join("%.4g" % (sum(d[c]) / len(d[c])) if c in d else "" for c in ctrs) + " |")
