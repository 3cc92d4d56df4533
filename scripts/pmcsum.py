"""Developer helper: average rocprofv3 --pmc counter values per dispatch of kernels whose name contains argv[2]."""
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, c), v in per.items():
        acc[c].append(v)
for c in sorted(acc):
    v = acc[c]
    print("%-28s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
