# ms/step of the single-GPU step, the N > 1 launch sequence (forced 1-rank RCCL group) and its one-graph form, same box, alternating
run() { env $1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29513 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --steps 50 --warmup 10 --no-compare --no-loader --no-decode --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', 'ms/step', d['ms_per_step'], 'buckets', d['config'].get('gradient_buckets'), 'loss', d.get('final_loss'))"; }
for i in 1 2 3; do
for e in "NACF_X=1" "NACF_BENCH_FORCE_DIST=1" "NACF_BENCH_FORCE_DIST=1 NACF_DDP_GRAPH_COLLECTIVES=1" "NACF_BENCH_FORCE_DIST=1 NACF_DDP_STAGES=3"; do run "$e"; done
done
