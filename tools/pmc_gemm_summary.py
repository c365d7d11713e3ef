import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        if "gemm_f32" not in r["Kernel_Name"]: continue
        k = (r["Kernel_Name"][:60], r["Grid_Size"])
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print(k, {n: round(sum(v)/len(v)) for n, v in c.items()}, "n=", len(next(iter(c.values()))))
