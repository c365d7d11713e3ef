# rocprofv3 kernel stats of the training leg under two environments (one box): tools/prof_ab.sh "A=1 B=2" "C=3"  -> gpurun_out/s4/prof_<i>.csv
cd /tmp && export TMPDIR=/tmp
i=0
for envs in "$@"; do
  i=$((i+1))
  rm -rf /tmp/prof_$i
  env $envs rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$i -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 10 --no-decode --no-compare --no-cpu-baseline --no-loader > /tmp/prof_$i.json 2>/dev/null
  cp /tmp/prof_$i/b_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/s4/prof_$i.csv
  echo "== $envs" >> $GRAFT_REPO_ROOT/gpurun_out/s4/prof_ab.txt
  tail -1 /tmp/prof_$i.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'])" >> $GRAFT_REPO_ROOT/gpurun_out/s4/prof_ab.txt
done
