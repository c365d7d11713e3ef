#!/bin/bash
# L1 / L2 counters of the exact-mode GEMM kernels on one shape per tile choice (tuning aid):
#   tools/pmc_gemm_cache2.sh "<gemm_bench --shapes spec>" "128 dma2 wide2"   -> gpurun_out/pmc_cache2.txt
set -u
SHAPES=${1:-0:5120:10547:512}
TILES=${2:-"128 dma2"}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_cache2.txt
cd /tmp && export TMPDIR=/tmp
: > $OUT
for t in $TILES; do
  echo "== tile $t shapes $SHAPES" >> $OUT
  i=0
  for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TA_BUSY_avr TD_BUSY_avr"; do
    i=$((i+1))
    rm -rf /tmp/pmc_c2_$i
    rocprofv3 --kernel-trace --output-format csv --pmc $grp -d /tmp/pmc_c2_$i -o r -- python $ROOT/tools/gemm_bench.py --iters 3 --modes bf16x3 --images --tiles $t --shapes $SHAPES > /tmp/pmc_c2_run.log 2>&1
    f=$(find /tmp/pmc_c2_$i -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python - "$f" >> $OUT <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "gemm_" in n and "wimage" not in n:
        agg[n[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print("  ", k, {n: "%.4g (x%d)" % (sum(v[-3:]) / len(v[-3:]), len(v)) for n, v in c.items()})
PY
    else tail -2 /tmp/pmc_c2_run.log >> $OUT; fi
  done
done
cat $OUT
