# A/B of environment settings inside ONE box: tools/ab_env.sh "A=1" "B=2" ... (each alternated 3 times; training step + NA decode)
run() { env $1 python bench.py --steps 50 --warmup 10 --no-compare --no-loader --no-cpu-baseline --decode-batches 6 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); dec=d.get('decode',{}); print('$1', 'ms/step', d['ms_per_step'], 'median', d.get('timing',{}).get('median_ms'), 'loss', d.get('final_loss'), 'decode ms/batch', dec.get('ms_per_batch'))"; }
for i in 1 2 3; do
for e in "$@"; do run "$e"; done
done
