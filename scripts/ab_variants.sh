#!/bin/bash
# A/B of library variants on the GPU box: for every pointclouds-rs_amd/lib/var_*.so (and the
# default library) one short bench run; prints step time, search-kernel time and the stage counts.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for lib in default pointclouds-rs_amd/lib/var_*.so; do
    [ -e "$lib" ] || [ "$lib" = default ] || continue
    if [ "$lib" = default ]; then unset PCRS_HIP_LIB; else export PCRS_HIP_LIB="$R/$lib"; fi
    timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-stages --icp-iters 0 "$@" > /tmp/ab.json 2> /tmp/ab.err || { echo "$lib failed"; tail -3 /tmp/ab.err; exit 1; }
    python3 -c "
import json,sys
d=json.load(open('/tmp/ab.json'))
print('$lib', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms_avg'],4), d['config']['search_stats_rank0'], d['config']['n_keep_rank0'])"
done
