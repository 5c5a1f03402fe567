export PCRS_GRID_MAXDIM=4096
for cfg in "0.75 8192 8192" "1.5 32768 8192" "1.5 32768 65536" "3.0 65536 65536"; do
  set -- $cfg
  export PCRS_GRID_REGION_MS=$1 PCRS_GRID_REGION_HEAVY=$2 PCRS_GRID_COARSE_MIN=$3
  echo "== region_ms $1 heavy $2 coarse_min $3"
  for s in 1 4 10 41.5; do timeout -k 10 280 python scripts/exp_aerial_big.py $s 2>&1 | grep points; done
  timeout -k 10 100 python scripts/exp_rings_loop.py rings dev 2>&1 | grep rings
  timeout -k 10 100 python scripts/exp_rings_loop.py blob dev 2>&1 | grep blob
done
