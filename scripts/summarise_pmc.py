"""Per-kernel sums of every counter collected by scripts/profile_r2_pmc.sh -> one small JSON kept under profiles/."""
import collections
import csv
import glob
import json
import sys

src, out = sys.argv[1], sys.argv[2]


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(src + "/g*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = short(row.get("Kernel_Name", "?"))
        c = row.get("Counter_Name")
        acc[k][c] += float(row.get("Counter_Value", 0))
        disp[k][c] += 1
res = {}
for k in acc:
    res[k] = {c: acc[k][c] for c in acc[k]}
    res[k]["dispatches"] = max(disp[k].values())
    h, m = acc[k].get("TCC_HIT_sum"), acc[k].get("TCC_MISS_sum")
    if h is not None and m is not None and h + m > 0:
        res[k]["l2_hit_rate"] = h / (h + m)
json.dump({"workload": "bench.py --config C2 --steps 1 (200 images), one launch sequence at a time", "kernels": res},
          open(out, "w"), indent=1, sort_keys=True)
print(json.dumps({k: res[k] for k in res if "describe" in k or "blur" in k or "nld" in k or "det_max" in k}, indent=1)[:6000])
