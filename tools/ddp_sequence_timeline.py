"""One step of the N > 1 launch sequence (staged backward, bucketed all-reduce on RCCL's stream, SyncBN exchanges) from a rocprofv3
--kernel-trace csv of `NACF_BENCH_FORCE_DIST=1 bench.py` (one GPU, a forced 1-rank RCCL group): every launch with start offset,
duration and queue, and per gradient bucket when its collective was issued, what had to land before it and how much compute is left
behind it -- the overlap window the first real N-rank run has to fit its all-reduce into.   python tools/ddp_sequence_timeline.py trace.csv"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step begins with rng_advance_kernel (once per step; the N > 1 sequence has one Adam walk per bucket)
first = [i for i, r in enumerate(rows) if "rng_advance_kernel" in r["Kernel_Name"]]
# (bench.py ends with rank-local profiling steps that are NOT the N > 1 sequence: take a step out of the timed region, a third in)
k = max(1, len(first) // 3)
a, b = first[k], first[k + 1]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
tend = int(step[-1]["End_Timestamp"])
def nm(r):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    return re.sub(r"\(.*", "", n)[:80]
is_coll = lambda r: "nccl" in r["Kernel_Name"].lower() or "rccl" in r["Kernel_Name"].lower()
print("# one step of the N > 1 sequence on one GPU (1-rank RCCL group): %d launches, span %.1f us" % (len(step), (tend - t0) / 1e3))
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  %7.1f us  queue %-3s %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), "** " if is_coll(r) else "", nm(r)))
colls = [r for r in step if is_coll(r)]
print("# collective kernels in the trace: %d (a 1-rank group reduces nothing: torch.distributed issues no kernel; the points where the"
      " N-rank run issues them are the ends of the backward stages, marked below)" % len(colls))
# bucket boundaries: a backward stage ends with its grouped weight-gradient launch + combine; its bucket is complete there
t_prev = t0
stage_ends = [r for r in step if "dw_group_reduce_kernel" in r["Kernel_Name"]]
adams = [r for r in step if "adam_step_kernel" in r["Kernel_Name"]]
for i, r in enumerate(stage_ends):
    e = int(r["End_Timestamp"])
    nxt = int(adams[0]["Start_Timestamp"]) if adams else tend
    left = sum(int(q["End_Timestamp"]) - int(q["Start_Timestamp"]) for q in step if int(q["Start_Timestamp"]) >= e and "adam" not in q["Kernel_Name"])
    print("#   backward stage %d complete (its gradient bucket can leave) at %.1f us; kernel time launched after it and before the first Adam walk: %.1f us"
          % (i, (e - t0) / 1e3, left / 1e3))
for i, r in enumerate(adams):
    print("#   Adam walk %d: %.1f .. %.1f us" % (i, (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3))
comp = [r for r in step if not is_coll(r)]
for i, c in enumerate(colls):
    s = int(c["Start_Timestamp"])
    left = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in comp if int(r["Start_Timestamp"]) >= s and "adam" not in r["Kernel_Name"])
    print("#   collective %d: starts at %.1f us, lasts %.1f us here (1 rank); compute launched after it and before Adam: %.1f us; step end at %.1f us"
          % (i, (s - t0) / 1e3, (int(c["End_Timestamp"]) - s) / 1e3, left / 1e3, (tend - t0) / 1e3))
