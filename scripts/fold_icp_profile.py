#!/usr/bin/env python3
"""Folds gpurun_out/prof_icp_<tag>/ (scripts/profile_icp.sh) into
profiles/<tag>_icp_kernel_stats.csv: per kernel its calls, total and mean duration and the median
duration over its launches (the mean of the ICP kernels is pulled up by the first iteration of
every registration, where most queries need a larger cube)."""
import collections, csv, glob, os, sqlite3, statistics, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
db = glob.glob(os.path.join(ROOT, "gpurun_out", "prof_icp_" + tag, "trace", "**", "*_results.db"),
               recursive=True)[0]
con = sqlite3.connect(db)
dur = collections.defaultdict(list)
for name, start, end in con.execute("select name,start,end from kernels order by start"):
    dur[name].append(end - start)
rows = sorted(dur.items(), key=lambda kv: -sum(kv[1]))
out = os.path.join(ROOT, "profiles", tag + "_icp_kernel_stats.csv")
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MedianNs"])
    for name, d in rows:
        w.writerow([name, len(d), sum(d), round(sum(d) / len(d), 1), int(statistics.median(d))])
print("wrote", out)
