"""Idle time between kernels of the steady-state training step, from a rocprofv3 --kernel-trace csv (tuning aid).
usage: python tools/trace_gaps.py <kernel_trace.csv> [n_last_steps]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
# steps end with the Adam kernel
ends = [i for i, e in enumerate(ev) if "adam_step_kernel" in e[2]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
tot_busy = tot_span = 0
gaps = {}
for a, b in zip(ends[-n - 1:-1], ends[-n:]):
    seg = ev[a + 1:b + 1]
    span = seg[-1][1] - seg[0][0]
    busy = sum(e[1] - e[0] for e in seg)
    tot_busy += busy; tot_span += span
    for x, y in zip(seg[:-1], seg[1:]):
        g = y[0] - x[1]
        k = (x[2][:50], y[2][:50])
        d = gaps.setdefault(k, [0, 0]); d[0] += g; d[1] += 1
print("steps %d: span %.3f ms  busy %.3f ms  idle %.3f ms (%.1f%%), %d kernels/step" % (
    n, tot_span / n / 1e6, tot_busy / n / 1e6, (tot_span - tot_busy) / n / 1e6, 100.0 * (tot_span - tot_busy) / tot_span, (ends[-1] - ends[-2])))
for k, (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    print("  %8.1f us/step (%5.2f us x %d)  %s -> %s" % (g / n / 1e3, g / c / 1e3, c // n, k[0], k[1]))
