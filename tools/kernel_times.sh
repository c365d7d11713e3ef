# average kernel time (rocprofv3 --kernel-trace --stats) of the kernels whose name contains one of the given words, inside the captured
# NACF step and the NA decode loop under the current environment:  tools/kernel_times.sh rowset argmax_merge ...
cd /tmp; export TMPDIR=/tmp
for leg in step decode; do
  rm -rf /tmp/pk
  if [ $leg = step ]; then c="python $GRAFT_REPO_ROOT/tools/step_profile.py 60"; else c="python $GRAFT_REPO_ROOT/tools/decode_profile.py 12"; fi
  METHOD=${METHOD:-NACF} BATCH=${BATCH:-128} MODE=${MODE:-bf16x3} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o b -- $c > /dev/null 2>&1
  echo "== $leg"
  python - "$@" <<PY
import csv, glob, sys
f = glob.glob('/tmp/pk/**/b_kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in sys.argv[1:]):
        print('   %-64s calls %5s  avg %8.2f us' % (r['Name'][:64], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
