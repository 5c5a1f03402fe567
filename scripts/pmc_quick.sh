#!/bin/bash
# One counter pass over a short run of bench.py: pmc_quick.sh <tag> <counters...>; results land
# in gpurun_out/pmc_<tag>/ and the per-kernel means are printed for kernels matching $PCRS_PMC_FILTER.
set -o pipefail
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/pmc_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc "$@" -d "$out/pmc" -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-stages --icp-iters 0 > "$out/run.log" 2>&1 || { tail -5 "$out/run.log"; exit 1; }
python3 $R/scripts/pmc_summary.py "$out/pmc" > "$out/summary.json"
python3 - "$out/summary.json" <<'PY'
import json,sys,os
d=json.load(open(sys.argv[1]))
flt=os.environ.get("PCRS_PMC_FILTER","knn_")
for k,v in d.items():
    if flt in k:
        print(k[:70], {c:round(x["mean_per_launch"]) for c,x in v.items()}, "launches", list(v.values())[0]["launches"])
PY
