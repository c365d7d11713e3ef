#!/bin/bash
# Collect everything profiles/ holds for one round, on the GPU box:  tools/collect_profiles.sh r02
# (run through gpurun; results land in gpurun_out/profiles_<round>/ and are then copied into profiles/ by hand)
set -u
R=${1:-r03}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/profiles_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the bench line itself (default flags)
python $ROOT/bench.py > $OUT/${R}_bench_line.json 2> $OUT/bench.err      # the driver's (compact) line
cp $ROOT/gpurun_out/bench_full.json $OUT/${R}_bench.json                      # the full record of the same run
# 2. rocprofv3 kernel stats of the same command
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o b -- python $ROOT/bench.py > /dev/null 2> /dev/null
cp $ROOT/gpurun_out/bench_full.json $OUT/${R}_bench_under_rocprof.json
cp /tmp/prof_stats/b_kernel_stats.csv $OUT/${R}_bench_kernel_stats.csv
# 2b. the training leg alone: every launch of a kernel class then has the shapes the bench's roofline object times
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o b -- python $ROOT/bench.py --no-decode --no-compare --no-cpu-baseline --no-loader > /dev/null 2> /dev/null
cp $ROOT/gpurun_out/bench_full.json $OUT/${R}_train_only_bench.json
cp /tmp/prof_train/b_kernel_stats.csv $OUT/${R}_train_only_kernel_stats.csv
# 3. HBM traffic counters, separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d /tmp/prof_$c -o b -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-compare --no-loader --graph off > /dev/null 2>&1
done
python $ROOT/tools/pmc_traffic.py /tmp/prof_FETCH_SIZE/b_counter_collection.csv /tmp/prof_WRITE_SIZE/b_counter_collection.csv $OUT/${R}_pmc_traffic.json train
# 3b. the NA-decode leg alone: kernel stats and its own HBM traffic table (a kernel's bytes depend on the launch's shape)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -o b -- python $ROOT/bench.py --decode-only > $OUT/${R}_decode_only_bench.json 2> /dev/null
cp /tmp/prof_dec/b_kernel_stats.csv $OUT/${R}_decode_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d /tmp/prof_dec_$c -o b -- python $ROOT/bench.py --decode-only --decode-batches 3 --graph off > /dev/null 2>&1
done
python $ROOT/tools/pmc_traffic.py /tmp/prof_dec_FETCH_SIZE/b_counter_collection.csv /tmp/prof_dec_WRITE_SIZE/b_counter_collection.csv $OUT/${R}_pmc_traffic_decode.json decode
# 4. GEMM microbenchmarks (three arithmetic modes, weights from pre-split images as in the model) + attainable MFMA peak
# (per-launch HIP events, median of 20; one-off stalls the tool saw go to the top of the file as '#' lines)
python $ROOT/tools/gemm_bench.py --iters 20 --modes f32,bf16x3,bf16 --images > /tmp/mb.txt 2> /tmp/mb.err
python $ROOT/tools/gemm_bench.py --iters 20 --modes bf16x3 --tiles 64,128,wide1,wide2,dma1,dma2,auto --images --only fwd >> /tmp/mb.txt 2>> /tmp/mb.err
python $ROOT/tools/gemm_bench.py --iters 20 --modes bf16x3 --tiles 64,128,wide1,wide2,dma1,dma2,auto --images --only dX >> /tmp/mb.txt 2>> /tmp/mb.err
{ echo "# per-launch HIP events, MEDIAN of 20 launches (tools/gemm_bench.py)"; grep "^#" /tmp/mb.err; cat /tmp/mb.txt; } > $OUT/${R}_gemm_microbench.txt
# 4b. the grouped weight-gradient launch per problem and as the step's mix
python $ROOT/tools/dw_group_bench.py 10 > $OUT/${R}_dw_group_bench.txt 2> /dev/null
# 4c. the same set on the 256 x 256 eight-phase body for fp32 operands against the 128 x 128 grouped kernel, both bf16 modes (graph replay)
{ for m in bf16 bf16x3; do echo "== mode $m"; MODE=$m python $ROOT/tools/dw_g256_bench.py 10 2>/dev/null | grep -v amdgpu.ids; done; } > $OUT/${R}_dw_g256_bench.txt
# 4d. the eight-phase bodies alone (standalone probes; built by hand, see their headers)
[ -x $ROOT/tools/probes/gemm256_8phase ] && $ROOT/tools/probes/gemm256_8phase 20 > $OUT/${R}_gemm256_8phase_probe.txt 2>&1
[ -x $ROOT/tools/probes/gemm256w_probe ] && $ROOT/tools/probes/gemm256w_probe 10 > $OUT/${R}_gemm256w_probe.txt 2>&1
# 4e. captured training steps of the throughput mode under rocprofv3 (configs[1] NAB B = 64, and NACF B = 128)
for cfg in "NACF 128 bf16 nacf128_bf16" "NAB 64 bf16 nab64_bf16"; do
  set -- $cfg
  rm -rf /tmp/prof_step
  METHOD=$1 BATCH=$2 MODE=$3 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_step -o b -- python $ROOT/tools/step_profile.py 300 > $OUT/${R}_step_$4.txt 2>/dev/null
  cp /tmp/prof_step/b_kernel_stats.csv $OUT/${R}_step_$4_kernel_stats.csv
done
$ROOT/tools/mfma_peak 20000 > $OUT/${R}_mfma_peak.txt 2>&1
# 5. SQ / GRBM counters of the vocabulary GEMM at K = 512 (the model's shape) and K = 8192, exact and throughput mode
{ for m in bf16x3 bf16; do echo "== mode $m"; $ROOT/tools/pmc_gemm.sh 0:5120:10547:512,0:5120:10547:8192 128 $m --images 2>/dev/null | grep "^pass"; done;
  echo "== mode bf16x3, wide kernel (128 x 256 tiles; csrc/gemm_bf16_wide.hpp)"; $ROOT/tools/pmc_gemm.sh 0:5120:10547:512,0:7680:512:2048,0:15360:1024:512 wide2 bf16x3 --images 2>/dev/null | grep "^pass"; } > $OUT/${R}_gemm_pmc_counters.txt
# 5a. the same counters on a decoder-layer shape (5120 x 512 x 512) for the 64 x 64 register-staged kernel and the DMA-fed kernels of
#     gemm_dma128.hpp ("no counter set exists for the 64 x 64 instantiation"), and the vocabulary shape on dma2
{ for t in 64 128 dma1 dma2; do echo "== mode bf16x3, tile $t, decoder-layer shapes"; $ROOT/tools/pmc_gemm.sh 0:5120:512:512,0:5120:2048:512,1:5120:512:2048 $t bf16x3 --images 2>/dev/null | grep "^pass"; done;
  echo "== mode bf16x3, tile dma2, vocabulary shape"; $ROOT/tools/pmc_gemm.sh 0:5120:10547:512,1:5120:10547:512 dma2 bf16x3 --images 2>/dev/null | grep "^pass"; } > $OUT/${R}_gemm_pmc_counters_decoder_shapes.txt
# 5a'. the DMA-fed kernel's phase stamps + ablations (tools/probes/dma128_probe.hip; built by hand, see its header)
for b in 1 2; do
  if [ -x $ROOT/tools/probes/dma128_probe_$b ]; then
    { echo "== dma128, MT=$b (workgroup tile $((64*b)) x 128)"; for shp in "5120 2048 512" "2944 2048 512" "5120 512 2048" "2304 10547 512" "15360 1024 512"; do $ROOT/tools/probes/dma128_probe_$b $shp 10; done; } >> $OUT/${R}_dma128_probe.txt 2>&1
  fi
done
# 5b. the wide kernel's phase stamps + ablations (tools/probes/wide_gemm.hip; built by hand, see its header)
for b in 1 2; do
  if [ -x $ROOT/tools/probes/wide_gemm$b ]; then
    { echo "== wide kernel, MT=$b (workgroup tile $((64*b)) x 256)"; for shp in "7680 512 2048" "15360 1024 512" "2304 10547 512" "15360 512 1024"; do $ROOT/tools/probes/wide_gemm$b $shp 10 | grep -v "by col\|by row\|entries off"; done; } >> $OUT/${R}_wide_gemm_probe.txt 2>&1
  fi
done
# 6. per-phase shader-clock cycles inside the bf16 kernels (needs the trace build: make -C .../csrc trace)
if [ -f $ROOT/non-autoregressive-video-captioning_amd/libnacf_hip_trace.so ]; then
  { for m in bf16x3 bf16; do for shp in 15360:1024:512 5120:10547:512; do echo "== mode $m shape $shp tile 128"; NACF_GEMM_MODE=$m python $ROOT/tools/bf16_trace.py $shp 128 --images 2>/dev/null; done; done; } > $OUT/${R}_bf16_phase_trace.txt
fi
ls -la $OUT
