# A/B of the non-GEMM tail changes inside ONE box (step times differ by ~4 % between boxes)
run() { env "$@" python bench.py --steps 50 --warmup 10 --no-compare --no-loader --no-decode --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d.get('timing',{}).get('median_ms'), d.get('final_loss'))"; }
for i in 1 2; do
run NACF_BN_MULTI=0 NACF_FUSED_ZERO_GRAD=0
run NACF_BN_MULTI=1 NACF_FUSED_ZERO_GRAD=0
run NACF_BN_MULTI=1 NACF_FUSED_ZERO_GRAD=1
done
