cd /root/repo
for rep in 1 2; do
for env in "" "HSA_ENABLE_INTERRUPT=0"; do
  echo "== env[$env] K=20"
  env $env python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.4g  ms_per_step %.5f  kernel_ms %.5f  overlapped %.5f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['overlapped']['ms_per_launch_overlapped']))"
done; done
echo "== default K"
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k: d[k] for k in ('value','ms_per_step','roofline')})[:900]); print(json.dumps(d.get('extras', {}))[:3000])"
