"""Condense the rocprofv3 CSV output of tools/profile_bench.sh into one JSON (per-kernel launch durations and counter sums)."""
import csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1]
out = {}


def rows(path):
    with open(path, newline="") as f:
        yield from csv.DictReader(f)


# kernel stats: <dir>/**/**_kernel_stats.csv (Name, Calls, TotalDurationNs, AverageNs, ...)
stats = []
for p in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in rows(p):
        stats.append({k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
out["kernel_stats"] = sorted(stats, key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))[:8]

for group in ("fetch", "write", "sq"):
    agg = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for p in glob.glob(os.path.join(root, group, "**", "*counter_collection.csv"), recursive=True):
        for r in rows(p):
            name = r.get("Kernel_Name") or r.get("Kernel Name") or "?"
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[name].add(r.get("Dispatch_Id"))
    out[group] = {k[:100]: {"launches": len(launches[k]), "counters_sum_over_launches": dict(v)}
                  for k, v in agg.items() if "kernel" in k}
print(json.dumps(out, indent=1))
