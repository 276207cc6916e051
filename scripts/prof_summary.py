"""Summarise a rocprofv3 --kernel-trace --output-format csv run: per-kernel calls / total / average (us)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    agg[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
tot = sum(sum(v) for v in agg.values())
print("%-70s %8s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%-70s %8d %12.1f %12.2f %6.1f%%" % (k[:70], len(v), sum(v), sum(v) / len(v), 100 * sum(v) / tot))
