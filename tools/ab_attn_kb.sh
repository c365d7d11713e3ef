# Kernel times of the attention family (and the criterion tail) inside the captured NACF step and the NA decode loop under several
# environments, one box:  tools/ab_attn_kb.sh "NACF_ATTN_KB=2" "NACF_ATTN_KB=1" "NACF_HIP_LIB=tools/ab/libnacf_hip_bf0.so" ...
# (each alternated twice; NACF_HIP_LIB = a tuning build of the library, e.g. -DNACF_ATTN_BRANCHFREE=0)   -> gpurun_out/attn_ab.txt
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/attn_ab.txt
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
: > $OUT
export TMPDIR=/tmp
for round in 1 2; do
for envs in "$@"; do
  for leg in step decode; do
    rm -rf /tmp/prof_kb
    if [ $leg = step ]; then cmd="python tools/step_profile.py 60"; else cmd="python tools/decode_profile.py 12"; fi
    env $envs METHOD=${METHOD:-NACF} BATCH=${BATCH:-128} MODE=${MODE:-bf16x3} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kb -o b -- $cmd > /tmp/prof_kb.log 2>&1
    echo "== $envs  [$leg]  $(grep -v rocprofv3 /tmp/prof_kb.log | tail -1)" >> $OUT
    python - <<PY >> $OUT
import csv, glob
f = glob.glob('/tmp/prof_kb/**/b_kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('crit_tail_fwd', 'attn::', 'lse_merge', 'attention')):
        print('   %-64s calls %5s  avg %8.2f us' % (r['Name'][:64], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  done
done
done
