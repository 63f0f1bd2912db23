"""Sum rocprofv3 --pmc counter CSVs per kernel: usage pmc_sum.py <dir> [kernel-substring]"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        if sub not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k)
    for c, v in sorted(acc[k].items()): print(f"   {c:40s} {v / cnt[(k, c)]:16.1f} per dispatch ({cnt[(k, c)]} dispatches)")
