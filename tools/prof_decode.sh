# rocprofv3 kernel stats of the NA decode loop (tools/decode_profile.py, B = 128): -> gpurun_out/s4/prof_decode.csv
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_dec
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -o b -- python $GRAFT_REPO_ROOT/tools/decode_profile.py 20 > $GRAFT_REPO_ROOT/gpurun_out/s4/prof_decode.txt 2>/dev/null
cp /tmp/prof_dec/b_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/s4/prof_decode.csv
