#!/bin/bash
# Kernel trace of one of the experiment scripts: bash scripts/profile_script.sh scripts/exp_x.py
# prints the total time per kernel over the whole run.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/prof_script
rm -rf "$out"; mkdir -p "$out"
script=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d "$out/trace" -o kt -- python3 "$script" "$@" > "$out/trace.log" 2>&1 || exit 1
python3 - "$out" <<'PY'
import sqlite3, sys, glob, collections
db = sqlite3.connect(glob.glob(sys.argv[1] + "/trace/**/*_results.db", recursive=True)[0])
tot = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in db.execute("select name,start,end from kernels"):
    k = n.split("(")[0][-44:]
    tot[k][0] += 1; tot[k][1] += (e - s) / 1000
for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {k:46s} x{c:5d} {t/1000:9.3f} ms total {t/c:9.1f} us each")
PY
tail -3 "$out/trace.log"
