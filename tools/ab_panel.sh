# A/B of the panel GEMM kernel inside ONE box: training step and NA decode, NACF_GEMM_PANEL=0 (off) / unset (by shape)
run() { env "$@" python bench.py --steps 50 --warmup 10 --no-compare --no-loader --no-cpu-baseline --decode-batches 6 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); dec=d.get('decode',{}); print('$*', 'ms/step', d['ms_per_step'], 'median', d.get('timing',{}).get('median_ms'), 'loss', d.get('final_loss'), 'decode', dec.get('value'), dec.get('ms_per_batch'))"; }
for i in 1 2 3; do
run NACF_GEMM_PANEL=0
run NACF_X=1
done
