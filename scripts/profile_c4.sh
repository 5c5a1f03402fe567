#!/bin/bash
# Kernel trace of the 100 M-point tile benchmark (bench.py --config c4): per-kernel time per step.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/prof_c4
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace -d "$out/trace" -o kt -- python3 $R/bench.py --config c4 --steps 3 --warmup 1 > "$out/trace.log" 2>&1 || exit 1
python3 - "$out" <<'PY'
import sqlite3, sys, glob, collections
db = sqlite3.connect(glob.glob(sys.argv[1] + "/trace/**/*_results.db", recursive=True)[0])
tot = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in db.execute("select name,start,end from kernels"):
    k = n.split("(")[0][-44:]
    tot[k][0] += 1; tot[k][1] += (e - s) / 1000
steps = 5  # one untimed pass for the search statistics, one warm-up, three timed
busy = sum(v[1] for v in tot.values())
print(f"kernels per step: {sum(v[0] for v in tot.values())/steps:.0f}, GPU busy per step {busy/steps/1000:.2f} ms")
for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"  {k:46s} x{c/steps:5.1f} {t/steps/1000:8.3f} ms/step")
PY
grep -o '"ms_per_step": [0-9.]*' "$out/trace.log"
