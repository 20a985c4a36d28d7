"""Per-kernel mean of every counter in a rocprofv3 --pmc CSV directory.
    python tools/pmc_kernels.py <dir> [substring ...]"""
import csv, glob, sys, collections
d = sys.argv[1]
subs = sys.argv[2:]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("syg::(anonymous namespace)::", "").split("(")[0]
    key = (name, r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))
    if subs and not any(s in name for s in subs):
        continue
    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, c in acc.items():
    print(key[0][:60], "grid", key[1], "lds", key[2], "n", len(next(iter(c.values()))))
    for k, v in sorted(c.items()):
        print("   %-28s %14.0f" % (k, sum(v) / len(v)))
