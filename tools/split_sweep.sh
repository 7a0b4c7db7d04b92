# usage: split_sweep.sh "<batches>:<parts,streams>" ...   (each run: bench.py --steps 6, no CPU baseline / host leg)
for item in "$@"; do
  b=${item%%:*}; cfg=${item##*:}
  KZG355_SPLIT=$cfg python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --batches-per-step $b 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$b batches, split $cfg, GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-default}:', round(d['value']), d['config']['step_ms'])"
done
