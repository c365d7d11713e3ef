# kernel trace of the N > 1 launch sequence on ONE GPU (forced 1-rank RCCL group): tools/ddp_sequence_trace.sh  -> gpurun_out/ddp_seq_kernel_trace.csv
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ddp
NACF_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ddp -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-decode --no-compare --no-loader --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/ddp_seq_bench.txt 2>&1
f=$(find /tmp/prof_ddp -name "b_kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/ddp_sequence_timeline.py $f > $GRAFT_REPO_ROOT/gpurun_out/ddp_sequence_timeline.txt
tail -2 $GRAFT_REPO_ROOT/gpurun_out/ddp_seq_bench.txt | cut -c1-600
