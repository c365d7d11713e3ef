run() { env "$@" python bench.py --steps 6 --warmup 2 --no-compare --no-loader --no-decode --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['final_loss'], d['ms_per_step'])"; }
for i in 1 2 3 4; do
run NACF_X=1
run NACF_BENCH_FORCE_DIST=1 NACF_BENCH_SYNC_BN=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=2957$i
done
