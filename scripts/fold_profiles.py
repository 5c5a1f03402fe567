#!/usr/bin/env python3
"""Folds gpurun_out/prof_<tag>/ (see profile_bench.sh) into profiles/<tag>_bench_kernel_stats.csv
(per-kernel calls / total / average / share of the kernel trace) and
profiles/<tag>_pmc_counters.json (per-kernel counter means per launch)."""
import csv, glob, json, os, sqlite3, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
db = glob.glob(os.path.join(src, "trace", "**", "*_results.db"), recursive=True)[0]
con = sqlite3.connect(db)
with open(os.path.join(dst, tag + "_bench_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
    for name, calls, total_us, avg_us, pct in con.execute(
            "select name,total_calls,total_duration,average,percentage from top_kernels"):
        w.writerow([name, calls, int(total_us * 1000), round(avg_us * 1000, 1), round(pct, 2)])
dirs = sorted(glob.glob(os.path.join(src, "pmc_*")))
dirs = [d for d in dirs if os.path.isdir(d)]
out = json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "pmc_summary.py")] + dirs))
# which state of the kernels the counters belong to, and the dominant kernel's duration in the
# kernel trace of the same command (the bench prices the counters with it)
commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()
dirty = subprocess.call(["git", "-C", ROOT, "diff", "--quiet", "--", "pointclouds-rs_amd/csrc"]) != 0
kernel_ms = None
for name, calls, total_us, avg_us, pct in con.execute(
        "select name,total_calls,total_duration,average,percentage from top_kernels"):
    if "knn_tag_kernel<17>" in name:
        kernel_ms = avg_us / 1000.0
out["_meta"] = {"commit": commit + ("+uncommitted kernel changes" if dirty else ""), "kernel_ms": kernel_ms,
                "command": "bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-stages --icp-iters 0"}
with open(os.path.join(dst, tag + "_pmc_counters.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote", dst)
