"""Per-step summary of a rocprofv3 --kernel-trace --stats csv of the training leg (bench.py --no-decode ...): microseconds and
launches per step by kernel and by class.  usage: kernel_stats_summary.py stats.csv [min_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
steps = max(int(r['Calls']) for r in rows if 'adam_step_kernel' in r['Name'])
tot, cls = 0.0, {}
for r in rows:
    n = r['Name']; t = float(r['TotalDurationNs']) / steps / 1e3; c = int(r['Calls']) / steps
    tot += t
    k = 'gemm' if 'gemm' in n else ('attention' if 'attn::' in n else ('aten' if ('at::native' in n or 'rocclr' in n) else 'other'))
    cls.setdefault(k, [0.0, 0.0]); cls[k][0] += t; cls[k][1] += c
    if t >= min_us:
        print("%7.1f us  x%4.1f  %s" % (t, c, n[:120]))
print("steps %d, kernel time per step %.1f us" % (steps, tot))
for k, (t, c) in sorted(cls.items(), key=lambda kv: -kv[1][0]):
    print("  %-10s %7.1f us  %5.1f launches" % (k, t, c))
