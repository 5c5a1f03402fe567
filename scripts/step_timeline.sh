#!/bin/bash
# Timeline of the kernels of one bench step on the GPU box: scripts/step_timeline.sh <tag> [bench args...]
# Leaves gpurun_out/prof_<tag>/timeline.txt: for the last step, every kernel's start (relative to the
# step's first kernel), duration and the idle time since the previous kernel ended.
set -o pipefail
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d "$out/trace" -o kt -- python3 "$R/bench.py" --steps 4 --warmup 2 --no-stages --no-cpu-baseline --icp-iters 0 "$@" > "$out/run.log" 2>&1 || { tail -5 "$out/run.log"; exit 1; }
python3 - "$out" <<'PY'
import glob, os, sqlite3, sys
out = sys.argv[1]
db = glob.glob(os.path.join(out, "trace", "**", "*_results.db"), recursive=True)[0]
con = sqlite3.connect(db)
rows = list(con.execute("select name, start, end from kernels order by start"))
# a step starts with the bounding-box kernel of its index build
starts = [i for i, r in enumerate(rows) if "bbox_kernel" in r[0]]
a = starts[-1]
step = rows[a:]
prev_begin = rows[starts[-2]][1] if len(starts) > 1 else None
with open(os.path.join(out, "timeline.txt"), "w") as f:
    if prev_begin is not None:
        f.write(f"step period (first kernel to first kernel): {(step[0][1] - prev_begin) / 1000:.1f} us\n")
    t0, last_end, busy = step[0][1], step[0][1], 0
    for name, s, e in step:
        f.write(f"{(s - t0) / 1000:8.1f} us  +{(e - s) / 1000:7.1f}  idle {(s - last_end) / 1000:6.1f}  {name[:90]}\n")
        busy += e - s
        last_end = max(last_end, e)
    f.write(f"first start to last end {(last_end - t0) / 1000:.1f} us, kernels {busy / 1000:.1f} us\n")
PY
tail -2 "$out/run.log" | cut -c1-300
cat "$out/timeline.txt"
