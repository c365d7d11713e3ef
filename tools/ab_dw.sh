# A/B of the grouped weight-gradient launch inside ONE box: 128x128 group kernel | all on the one-workgroup-per-CU kernel (NACF_DW_WIDE=1) | split by row list (=2)
run() { env "$@" python bench.py --steps 50 --warmup 10 --no-compare --no-loader --no-decode --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$*: step', d['ms_per_step'], 'ms (median', d.get('timing',{}).get('median_ms'), '); dominant', r['kernel'][:40], r['achieved'], 'TF', r.get('avg_launch_ms'), 'ms; loss', d.get('final_loss'))"; }
run NACF_DW_WIDE=0
run NACF_DW_WIDE=2
run NACF_DW_WIDE=2 NACF_DW_GROUP_WGS=1024
run NACF_DW_WIDE=2 NACF_DW_GROUP_WGS=640
run NACF_DW_WIDE=1 NACF_DW_GROUP_WGS=512
run NACF_DW_WIDE=0
