#!/bin/bash
# Kernel trace of the frame pipeline benchmark (bench.py --config c5): prints per-kernel totals
# per frame and the share of the frame time the GPU is busy.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/prof_c5
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d "$out/trace" -o kt -- python3 $R/bench.py --config c5 --steps 20 --warmup 3 > "$out/trace.log" 2>&1 || exit 1
python3 - "$out" <<'PY'
import sqlite3, sys, glob, collections
db = sqlite3.connect(glob.glob(sys.argv[1] + "/trace/**/*_results.db", recursive=True)[0])
tot = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in db.execute("select name,start,end from kernels"):
    k = n.split("(")[0][-44:]
    tot[k][0] += 1; tot[k][1] += (e - s) / 1000
frames = 23
busy = sum(v[1] for v in tot.values())
print(f"kernels per frame: {sum(v[0] for v in tot.values())/frames:.0f}, GPU busy per frame {busy/frames:.0f} us")
for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {k:46s} x{c/frames:5.1f} {t/frames:7.1f} us/frame")
try:
    rows = list(db.execute("select count(*), sum(end-start) from memory_copies"))
    print("memory copies per frame:", rows[0][0] / frames, "us/frame:", rows[0][1] / 1000 / frames)
except Exception as ex:
    print("no copy table", ex)
PY
grep -o '"ms_per_step": [0-9.]*' "$out/trace.log"
