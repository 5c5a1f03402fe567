#!/bin/bash
# Kernel trace of any script of this repository on the GPU box: scripts/profile_any.sh <tag> <script> [args...]
# Leaves gpurun_out/prof_<tag>/kernel_stats.txt (per-kernel calls / total / average of the trace).
set -o pipefail
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/trace" -o kt -- python3 "$R/$1" "${@:2}" > "$out/run.log" 2>&1 || { tail -5 "$out/run.log"; exit 1; }
python3 - "$out" <<'PY'
import glob, os, sqlite3, sys
out = sys.argv[1]
db = glob.glob(os.path.join(out, "trace", "**", "*_results.db"), recursive=True)[0]
con = sqlite3.connect(db)
with open(os.path.join(out, "kernel_stats.txt"), "w") as f:
    for name, calls, total_us, avg_us, pct in con.execute(
            "select name,total_calls,total_duration,average,percentage from top_kernels"):
        f.write(f"{avg_us:10.1f} us x {calls:6d} = {total_us/1000:9.2f} ms {pct:6.2f}%  {name[:110]}\n")
# the last launches in the order they ran (start relative to the first of them, duration, idle time before)
rows = list(con.execute("select name, start, end from kernels order by start"))[-int(os.environ.get("TIMELINE_LAST", "90")):]
with open(os.path.join(out, "timeline.txt"), "w") as f:
    t0 = last = rows[0][1]
    for name, s, e in rows:
        f.write(f"{(s - t0) / 1000:9.1f} us  +{(e - s) / 1000:8.1f}  idle {(s - last) / 1000:7.1f}  {name[:90]}\n")
        last = max(last, e)
PY
tail -3 "$out/run.log"
