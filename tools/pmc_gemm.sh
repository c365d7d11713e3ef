#!/bin/bash
# SQ / GRBM counters of the GEMM kernels (tuning aid): tools/pmc_gemm.sh "<gemm_bench --shapes spec>" [tile] [mode] [--images]
# writes gpurun_out/pmc_gemm/*.csv and prints per-kernel averages (tools/pmc_gemm_summary.py)
set -u
SHAPES=${1:-0:5120:10547:512}
TILE=${2:-128}
MODE=${3:-f32}
IMAGES=${4:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_gemm
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_MFMA"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d /tmp/pmc_gemm_$i -o r -- python $ROOT/tools/gemm_bench.py --iters 5 --tiles $TILE --modes $MODE $IMAGES --shapes $SHAPES > $OUT/run_$i.log 2>&1
  tail -3 $OUT/run_$i.log
  f=$(find /tmp/pmc_gemm_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/counters_$i.csv
  t=$(find /tmp/pmc_gemm_$i -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && cp $t $OUT/trace_$i.csv
done
python $ROOT/tools/pmc_gemm_summary.py $OUT
