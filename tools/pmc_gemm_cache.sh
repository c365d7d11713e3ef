#!/bin/bash
# L1 / L2 counters of the GEMM kernels (tuning aid): tools/pmc_gemm_cache.sh "<gemm_bench --shapes spec>" [tile]
set -u
SHAPES=${1:-0:5120:10547:512}
export NACF_GEMM_TILE=${2:-128}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_cache
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TA_BUSY_avr TD_BUSY_avr" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d /tmp/pmc_cache_$i -o r -- python $ROOT/tools/gemm_bench.py --iters 3 --tiles $NACF_GEMM_TILE --shapes $SHAPES > $OUT/run_$i.log 2>&1
  f=$(find /tmp/pmc_cache_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_f32" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][5:42]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print(k, {n: "%.4g" % (sum(v) / len(v)) for n, v in c.items()})
PY
  else tail -2 $OUT/run_$i.log; fi
done
