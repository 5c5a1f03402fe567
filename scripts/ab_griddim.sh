#!/bin/bash
# A/B of the grid sizing (PCRS_GRID_DIM=0: aim at the volume occupancy whatever the cloud is;
# default: the aim follows the dimension the occupancy reveals) over clouds of different shape.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for h in 1 0; do
    export PCRS_GRID_DIM=$h
    echo "== PCRS_GRID_DIM=$h"
    timeout -k 10 200 python3 scripts/exp_surfaces.py 2>&1 | grep -v "amdgpu.ids\|^pcrs grid" || exit 1
    timeout -k 10 120 python3 scripts/exp_lidar_rings.py 8 2>&1 | grep sweep || exit 1
    timeout -k 10 120 python3 scripts/exp_kitti_pipeline.py 2>&1 | grep "rep 2" || exit 1
    timeout -k 10 300 python3 scripts/exp_strays.py 2>&1 | tail -2 || exit 1
    timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-stages --icp-iters 0 | cut -c1-200
    timeout -k 10 300 python3 bench.py --config c4 --steps 3 --warmup 1 | cut -c1-200
done
