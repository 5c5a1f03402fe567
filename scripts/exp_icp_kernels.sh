#!/bin/bash
# Per-kernel durations of the steady-state ICP iteration (bench.py --config c3) from one kernel
# trace: median over the launches of each kernel.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/exp_icp_$1
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d "$out/trace" -o kt -- python3 $R/bench.py --config c3 > "$out/trace.log" 2>&1 || exit 1
python3 - "$out" <<'PY'
import sqlite3, sys, glob, statistics, collections
db = sqlite3.connect(glob.glob(sys.argv[1] + "/trace/**/*_results.db", recursive=True)[0])
seq = collections.defaultdict(list)
for n, s, e in db.execute("select name,start,end from kernels order by start"):
    seq[n.split("(")[0]].append((e - s) / 1000)
for k, v in seq.items():
    if len(v) >= 20:
        print(f"{k[-40:]:40s} n={len(v):4d} median_us={statistics.median(v):6.1f}")
PY
grep -o '"ms_per_step": [0-9.]*' "$out/trace.log"
