"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files (several passes merged)."""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(collections.Counter)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        k = k[k.find("sz::"):][:70] if "sz::" in k else k[:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        print("    %-28s %16.0f  (n=%d)" % (c, agg[k][c] / cnt[k][c], cnt[k][c]))
