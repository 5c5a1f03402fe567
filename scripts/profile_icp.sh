#!/bin/bash
# Kernel trace of the ICP benchmark (bench.py --config c3) on the GPU box; the summary lands in
# gpurun_out/prof_icp_<tag>/ (copy what is to be kept into profiles/).
set -o pipefail
tag=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/prof_icp_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/trace" -o kt -- python3 $R/bench.py --config c3 > "$out/trace.log" 2>&1 || exit 1
grep "^{" "$out/trace.log" | cut -c1-400
echo profiled
