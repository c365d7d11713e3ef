# decode kernel stats under two environments in one box: tools/prof_decode_ab.sh "A=1" "B=2"  -> gpurun_out/<tag>_decode_stats.txt
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for e in "$@"; do
  i=$((i+1))
  rm -rf /tmp/prof_dec
  env $e rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -o b -- python $GRAFT_REPO_ROOT/tools/decode_profile.py 20 > /dev/null 2>&1
  echo "== $e" >> $GRAFT_REPO_ROOT/gpurun_out/decode_ab_stats.txt
  python - <<PY >> $GRAFT_REPO_ROOT/gpurun_out/decode_ab_stats.txt
import csv,glob
rows=list(csv.DictReader(open(glob.glob('/tmp/prof_dec/**/b_kernel_stats.csv',recursive=True)[0])))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]:
    print("%8.1f us total  x%5d  avg %7.1f us  %5.1f%%  %s"%(float(r['TotalDurationNs'])/1e3,int(r['Calls']),float(r['AverageNs'])/1e3,100*float(r['TotalDurationNs'])/tot,r['Name'][:100]))
print("total %.1f us"%(tot/1e3))
PY
done
