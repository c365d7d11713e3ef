# rocprofv3 kernel stats of a captured step:  METHOD=NACF BATCH=128 MODE=bf16 tools/prof_step.sh <tag>   (results in gpurun_out/s5/)
TAG=${1:-step}
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/s5
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_step
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_step -o b -- python $GRAFT_REPO_ROOT/tools/step_profile.py 300 > $GRAFT_REPO_ROOT/gpurun_out/s5/${TAG}.txt 2>/dev/null
cp /tmp/prof_step/b_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/s5/${TAG}.csv
