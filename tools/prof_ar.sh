cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ar
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ar -o b -- python $GRAFT_REPO_ROOT/tools/ar_profile.py 8 > $GRAFT_REPO_ROOT/gpurun_out/s4/prof_ar.txt 2>/dev/null
cp /tmp/prof_ar/b_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/s4/prof_ar.csv
