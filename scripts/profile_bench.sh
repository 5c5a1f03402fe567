#!/bin/bash
# Profiles `bench.py` on the GPU box: one kernel trace and separate counter passes (rocprofv3
# cannot split counters over passes itself).  Results land under gpurun_out/prof_<tag>/ and are
# folded into profiles/ by scripts/fold_profiles.py.  Delete the local gpurun_out/prof_<tag>/ of an
# earlier run first: results merge into it, and the fold would average old and new launches.
set -o pipefail
tag=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
args="$R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-stages --icp-iters 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/trace" -o kt -- python3 $args > "$out/trace.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d "$out/pmc_fetch" -- python3 $args > "$out/pmc_fetch.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d "$out/pmc_write" -- python3 $args > "$out/pmc_write.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d "$out/pmc_sq1" -- python3 $args > "$out/pmc_sq1.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d "$out/pmc_sq2" -- python3 $args > "$out/pmc_sq2.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE -d "$out/pmc_ta" -- python3 $args > "$out/pmc_ta.log" 2>&1 || exit 1
grep "^{" "$out/trace.log" | cut -c1-300
echo profiled
