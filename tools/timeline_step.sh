# kernel timeline of ONE captured NACF training step under the current environment:  tools/timeline_step.sh <tag>   (gpurun_out/<tag>_timeline.txt)
TAG=${1:-step}
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tl
METHOD=${METHOD:-NACF} BATCH=${BATCH:-128} MODE=${MODE:-bf16x3} rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -o b -- python $GRAFT_REPO_ROOT/tools/step_profile.py 40 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/step_timeline.py $(find /tmp/prof_tl -name "b_kernel_trace.csv" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_timeline.txt
