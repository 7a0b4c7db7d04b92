for cf in default 1; do for n in 512 1024 2048 4096; do
  if [ $cf = default ]; then unset KZG355_LC_CHAIN_FROM; else export KZG355_LC_CHAIN_FROM=$cf; fi
  python bench.py --batches-per-step $n --pipeline 3 --steps 24 --warmup 6 --no-cpu-baseline --no-host-leg --no-latency 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chain_from=$cf batches=$n pipeline=3 value', round(d['value']), 'ms/step', round(d['ms_per_step'],2))"
  python bench.py --batches-per-step $n --steps 12 --warmup 3 --no-cpu-baseline --no-host-leg --no-latency 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chain_from=$cf batches=$n sync       value', round(d['value']), 'ms/step', round(d['ms_per_step'],2), {k: round(v.get('avg_launch_ms',0),3) for k,v in d['roofline']['per_kernel'].items() if k in ('lincomb_horner','pairing','lincomb')})"
done; done
