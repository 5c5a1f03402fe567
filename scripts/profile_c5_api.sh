#!/bin/bash
# HIP API trace of the frame pipeline benchmark (bench.py --config c5): copies and waits per frame.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/prof_c5_api
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --hip-trace -d "$out/trace" -o ht -- python3 $R/bench.py --config c5 --steps 40 --warmup 3 --no-stages > "$out/trace.log" 2>&1 || { tail -5 "$out/trace.log"; exit 1; }
python3 - "$out" <<'PY' | tee "$out/api_per_frame.txt"
import glob, sqlite3, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + "/trace/**/*_results.db", recursive=True)[0])
rows = db.execute("select name, count(*), avg(end - start) from regions group by name order by 2 desc").fetchall()
frames = 43  # warmup + steps, all through the one-call pipeline (--no-stages)
for name, c, avg in rows:
    if any(k in name for k in ("Memcpy", "Synchronize", "LaunchKernel", "Malloc", "Free", "Memset", "EventRecord")):
        print(f"  {name[:40]:42s} {c / frames:7.2f} per frame   {avg / 1e3:8.1f} us each")
PY
grep -o '"ms_per_step": [0-9.]*' "$out/trace.log"
