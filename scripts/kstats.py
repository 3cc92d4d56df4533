"""Pretty-print a rocprofv3 kernel_stats.csv (developer helper)."""
import csv, glob, sys
f = (glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    n = r["Name"].replace("bmx::(anonymous namespace)::", "").replace("void ", "")[:44]
    print("%-46s calls=%5s total_ms=%9.2f avg_ms=%8.3f" % (n, r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
print("total kernel ms", tot / 1e6)
