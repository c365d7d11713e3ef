#!/bin/bash
# SQ / GRBM counters of the grouped weight-gradient kernel on the NACF step's set (tools/dw_g256_bench.py), both bf16 modes:
#   tools/pmc_dw_g256.sh  ->  gpurun_out/pmc_dw_g256.txt   (counters-only passes: --kernel-trace + --pmc, one group per run)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_dw_g256.txt
cd /tmp && export TMPDIR=/tmp
: > $OUT
for mode in bf16x3 bf16; do
  echo "== mode $mode" >> $OUT
  i=0
  for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_MFMA"; do
    i=$((i+1))
    rm -rf /tmp/pmc_dw_$i
    MODE=$mode rocprofv3 --kernel-trace --output-format csv --pmc $grp -d /tmp/pmc_dw_$i -o r -- python $ROOT/tools/dw_g256_bench.py 3 > /dev/null 2>&1
    f=$(find /tmp/pmc_dw_$i -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python - "$f" >> $OUT <<'PY'
import csv, collections, sys
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "g256_dw_group_kernel" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    live = [x for x in v if x > 0] or v        # (warm-up launches of an empty set count nothing: not part of the average)
    print("  %-28s avg over %d launches  %.4g" % (k, len(live), sum(live) / len(live)))
PY
  done
done
# matrix-pipe share of the whole launch, tools/pmc_gemm_summary.py's convention: GRBM_GUI_ACTIVE is summed over the 8 XCDs
python - $OUT >> $OUT <<'PY'
import re, sys
mode, vals = None, {}
for line in open(sys.argv[1]):
    m = re.match(r"== mode (\S+)", line)
    if m:
        mode = m.group(1); vals[mode] = {}
        continue
    m = re.match(r"\s+(\S+)\s+avg over \d+ launches\s+(\S+)", line)
    if m and mode:
        vals[mode][m.group(1)] = float(m.group(2))
for mode, v in vals.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
        print("# %s: matrix pipe busy %.1f %% of the launch = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)"
              % (mode, 100.0 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)))
PY
cat $OUT
