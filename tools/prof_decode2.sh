# rocprofv3 kernel stats of the NA decode loop for a paradigm: tools/prof_decode2.sh ef|l2r [q]
cd /tmp && export TMPDIR=/tmp
P=${1:-ef}; Q=${2:-1}
rm -rf /tmp/prof_dec
DECODE_PARADIGM=$P DECODE_Q=$Q rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -o b -- python $GRAFT_REPO_ROOT/tools/decode_profile.py 6 > $GRAFT_REPO_ROOT/gpurun_out/s4/prof_decode_$P.txt 2>/dev/null
cp /tmp/prof_dec/b_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/s4/prof_decode_$P.csv
