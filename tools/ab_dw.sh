# A/B of the grouped weight-gradient launch inside ONE box: XCD order x workgroup target
for o in 1 0; do for w in 768 1024 1280 1792 2560; do
  NACF_DW_GROUP_ORDER=$o NACF_DW_GROUP_WGS=$w python bench.py --steps 50 --warmup 10 --no-compare --no-loader --no-decode --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('order $o target $w: step', d['ms_per_step'], 'ms; grouped dW', r['achieved'], 'TF', r.get('avg_launch_ms'), 'ms', d.get('final_loss'))"
done; done
