"""The kernel sequence of ONE captured training step from a rocprofv3 --kernel-trace csv:  python tools/step_timeline.py b_kernel_trace.csv
Takes the launches between the last two Adam walks; prints start offset, duration, grid, workgroup and the kernel name."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_step_kernel" in r["Kernel_Name"]]
a, b = adam[-2], adam[-1]
t0 = int(rows[a + 1]["Start_Timestamp"])
tot = 0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)[:90]
    grid = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) * max(1, int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"]))) * max(1, int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_Z"])))
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:7.1f} us  wgs {grid:6d} x {r['Workgroup_Size_X']:>4s}  {name}")
    tot += e - s
print(f"sum of kernel durations {tot / 1e3:.1f} us, span {(int(rows[b]['End_Timestamp']) - t0) / 1e3:.1f} us, {b - a} launches")
