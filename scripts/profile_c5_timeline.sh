#!/bin/bash
# Kernel + copy timeline of one frame of bench.py --config c5 (the one-call pipeline).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/prof_c5_timeline
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace -d "$out/trace" -o kt -- python3 $R/bench.py --config c5 --steps 10 --warmup 3 --no-stages > "$out/trace.log" 2>&1 || { tail -5 "$out/trace.log"; exit 1; }
python3 - "$out" <<'PY' > "$out/timeline.txt"
import glob, sqlite3, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + "/trace/**/*_results.db", recursive=True)[0])
ev = [(s, e, n.split("(")[0][-60:]) for n, s, e in db.execute("select name,start,end from kernels")]
try:
    ev += [(s, e, "COPY " + str(n)) for n, s, e in db.execute("select name,start,end from memory_copies")]
except Exception as ex:
    print("no copy table:", ex)
ev.sort()
# the last frame: from the last upload run back to the previous one
ups = [i for i, x in enumerate(ev) if x[2].startswith("COPY") and "HOST_TO_DEVICE" in x[2].upper()]
starts = [i for j, i in enumerate(ups) if j == 0 or ev[i][0] - ev[ups[j - 1]][0] > 300000]
a = starts[-1]
frame = ev[a:]
t0 = last = frame[0][0]
busy = 0
for s, e, n in frame:
    print(f"{(s - t0) / 1000:8.1f} us  +{(e - s) / 1000:7.1f}  idle {(s - last) / 1000:6.1f}  {n}")
    busy += e - s
    last = max(last, e)
print(f"frame: first start to last end {(last - t0) / 1000:.1f} us, busy {busy / 1000:.1f} us, {len(frame)} events")
PY
tail -n 3 "$out/timeline.txt"; grep -o '"ms_per_step": [0-9.]*' "$out/trace.log"
