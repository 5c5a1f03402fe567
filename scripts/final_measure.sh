#!/bin/bash
# The measurements the design notes quote, in one go on the GPU box (results under gpurun_out/<tag>/).
tag=${1:-final}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/$tag
mkdir -p "$out"
cd "$R"
run() { name=$1; shift; echo "== $name"; timeout -k 10 "$@" > "$out/$name.log" 2> "$out/$name.err" || { echo "FAILED $name"; tail -n 5 "$out/$name.err"; exit 1; }; }
run bench_c2 400 python bench.py
run bench_c3 300 python bench.py --config c3
run bench_c4 700 python bench.py --config c4
run bench_c5 300 python bench.py --config c5
run shard_slabs 300 python scripts/exp_shard_cost.py slabs
run shard_slabs_c4 400 python scripts/exp_shard_cost.py slabs 8 c4
run rings 300 python scripts/exp_lidar_rings.py 1 8
run rings_dev 200 python scripts/exp_rings_loop.py rings dev
run blob_dev 200 python scripts/exp_rings_loop.py blob dev
run clusters 200 python scripts/exp_clusters.py
run aerial_24m 280 python scripts/exp_aerial_big.py 10
run pathological 300 python scripts/exp_pathological.py
run stats 300 python scripts/exp_stats.py 1e6 8e6 1e8
run small_cloud 300 python scripts/exp_small_cloud.py
echo all done
