#!/bin/bash
# N>1 readiness check on any N-GPU node (the build box has ONE GPU: this script is for whoever has a node):
#   tools/scale_check.sh [N]            (default: all visible GPUs)
# Runs bench.py at 1 and N ranks (torch.distributed.run, RCCL), prints per-N videos/s, the scaling efficiency, the ranks
# RCCL saw, every rank's loss (must all be finite) and the exposed communication time per step
# (ms_per_step(N) - ms_per_step(1)); exits non-zero if a run fails, a loss is not finite, or the ranks' losses diverge.
set -u
cd "$(dirname "$0")/.."
N=${1:-$(python -c 'import torch; print(torch.cuda.device_count())')}
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=${OUT:-gpurun_out/scale_check}
mkdir -p $OUT
run() {   # $1 = ranks
  if [ "$1" = "1" ]; then
    python bench.py --gpus 1 --steps 50 --warmup 5 --no-compare --no-loader --no-decode --no-cpu-baseline > $OUT/n$1.log 2>&1
  else
    NCCL_DEBUG=${NCCL_DEBUG:-VERSION} python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 \
      --master-port $((29600 + $1)) bench.py --gpus $1 --steps 50 --warmup 5 --no-compare --no-loader > $OUT/n$1.log 2>&1
  fi
  tail -1 $OUT/n$1.log > $OUT/n$1.json
}
run 1 || { echo "1-rank run failed"; tail -20 $OUT/n1.log; exit 1; }
[ "$N" -gt 1 ] && { run $N || { echo "$N-rank run failed"; tail -30 $OUT/n$N.log; exit 1; }; }
python - "$OUT" "$N" <<'PY'
import json, math, sys
out, n = sys.argv[1], int(sys.argv[2])
one = json.load(open("%s/n1.json" % out))
print("1 rank : %9.1f videos/s  %.3f ms/step  (gemm mode %s)" % (one["value"], one["ms_per_step"], one["config"]["gemm_mode"]))
if n > 1:
    many = json.load(open("%s/n%d.json" % (out, n)))
    c = many["config"]
    print("%d ranks: %9.1f videos/s  %.3f ms/step  efficiency %.3f  exposed comm+launch %.3f ms/step" % (
        n, many["value"], many["ms_per_step"], many["value"] / (n * one["value"]), many["ms_per_step"] - one["ms_per_step"]))
    print("         ranks seen by RCCL: %d  buckets %s  overlapped %s  sync_bn %s  hipgraph %s" % (
        many["n_gpus"], c["gradient_buckets"], c["overlapped_allreduce"], c["sync_bn"], c["hipgraph"]))
    losses = many.get("rank_losses") or []
    print("         per-rank loss:", losses)
    assert many["n_gpus"] == n and len(losses) == n, "RCCL did not see %d ranks" % n
    assert all(math.isfinite(x) for x in losses), "non-finite loss on a rank"
    # different shards, same weights: the losses differ by sampling noise only
    assert max(losses) - min(losses) < 0.25 * abs(sum(losses) / n), "rank losses diverge"
    print("OK")
PY
# the same N-rank run with the RCCL calls captured inside ONE step graph (runtime/engine.py, NACF_DDP_GRAPH_COLLECTIVES=1):
# at one rank it halves the launch overhead of the N > 1 sequence; it has never run with N > 1 -- this is the run that tells
if [ "$N" -gt 1 ]; then
  NACF_DDP_GRAPH_COLLECTIVES=1 NCCL_DEBUG=${NCCL_DEBUG:-VERSION} timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
    --master-addr 127.0.0.1 --master-port $((29700 + $N)) bench.py --gpus $N --steps 50 --warmup 5 --no-compare --no-loader > $OUT/n${N}_graph.log 2>&1 \
    && tail -1 $OUT/n${N}_graph.log | python -c "
import json, sys
d = json.loads(sys.stdin.read()); one = json.load(open('$OUT/n1.json'))
print('%d ranks, collectives inside the step graph: %9.1f videos/s  %.3f ms/step  efficiency %.3f  losses %s' % (
    d['n_gpus'], d['value'], d['ms_per_step'], d['value'] / (d['n_gpus'] * one['value']), d.get('rank_losses')))" \
    || { echo "graph-captured collectives FAILED at $N ranks (keep NACF_DDP_GRAPH_COLLECTIVES=0):"; tail -15 $OUT/n${N}_graph.log; }
fi
