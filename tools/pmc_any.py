"""Per-kernel averages of arbitrary rocprofv3 --pmc counters: python tools/pmc_any.py <counter_collection.csv> [more.csv ...]"""
import collections, csv, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0][:44] + " g=%s" % r["Grid_Size"]
        a = agg[k][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        n, v = agg[k][c]
        print("    %-30s %16.1f  (n=%d)" % (c, v / n, n))
