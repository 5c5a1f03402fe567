#!/bin/bash
# A/B of the two lane-per-query search kernels that return neighbour identities
# (PCRS_ONE_WALK=0: two walks, knn_cell_kernel; default: one walk with tagged keys, knn_tag_kernel)
# over the experiment scripts that stress them differently.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for h in 1 0; do
    export PCRS_ONE_WALK=$h
    echo "== PCRS_ONE_WALK=$h"
    timeout -k 10 120 python3 scripts/exp_kernel_time.py 1000000 || exit 1
    timeout -k 10 120 python3 scripts/exp_lidar_rings.py 8 || exit 1
    timeout -k 10 120 python3 scripts/exp_pathological.py || exit 1
    timeout -k 10 120 python3 scripts/exp_kitti_pipeline.py || exit 1
    timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-stages --icp-iters 0 | cut -c1-200
done
