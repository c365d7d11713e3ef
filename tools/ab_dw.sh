set -x
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "dw or group" 2>&1 | tail -3
for o in 0 1 0 1; do
  NACF_DW_GROUP_ORDER=$o python bench.py --steps 50 --warmup 10 --no-compare --no-loader --no-decode --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('order $o', d['ms_per_step'], r.get('kernel'), r['achieved'], r['frac'], d.get('final_loss'))"
done
