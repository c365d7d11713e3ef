"""Per-dispatch SQ / GRBM counters of the GEMM kernels collected by tools/pmc_gemm.sh (tuning aid).
usage: python tools/pmc_gemm_summary.py <dir with counters_N.csv + trace_N.csv>
Derived: effective clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs
over GRBM_GUI_ACTIVE / 8 (fraction of active cycles in which a SIMD's matrix pipe is busy)."""
import collections
import csv
import glob
import os
import sys


def main():
    base = sys.argv[1]
    per = collections.defaultdict(dict)     # (pass, dispatch) -> counters
    for path in sorted(glob.glob(os.path.join(base, "counters_*.csv"))):
        i = os.path.basename(path).split("_")[1].split(".")[0]
        dur = {}
        for r in csv.DictReader(open(os.path.join(base, "trace_%s.csv" % i))):
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for r in csv.DictReader(open(path)):
            if not any(k in r["Kernel_Name"] for k in ("gemm_f32_kernel", "gemm_bf16_kernel", "gemm_wide_kernel")):
                continue
            d = per[(i, r["Dispatch_Id"])]
            d["kernel"] = r["Kernel_Name"].split("(")[0][5:64]
            d["grid"] = r["Grid_Size"]
            d["us"] = dur.get(r["Dispatch_Id"], 0) / 1e3
            d[r["Counter_Name"]] = float(r["Counter_Value"])
    seen = collections.Counter()
    for (i, disp), d in sorted(per.items(), key=lambda kv: (kv[0][0], int(kv[0][1]))):
        seen[(i, d["kernel"])] += 1
        if seen[(i, d["kernel"])] > 2 and "--all" not in sys.argv:      # two dispatches per kernel and pass are enough
            continue
        extra = ""
        if "GRBM_GUI_ACTIVE" in d and d["us"] > 0:
            act = d["GRBM_GUI_ACTIVE"] / 8.0
            extra = " clock %.2f GHz  MFMA-busy %.1f%%" % (act / (d["us"] * 1e3), 100.0 * d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0 / act)
        if "SQ_WAIT_ANY" in d:
            tot = d["SQ_WAIT_ANY"] + d["SQ_WAIT_INST_ANY"] + d["SQ_ACTIVE_INST_ANY"]
            extra = " wave cycles: parked %.1f%%  issue-stall %.1f%%  active %.1f%%" % (
                100 * d["SQ_WAIT_ANY"] / tot, 100 * d["SQ_WAIT_INST_ANY"] / tot, 100 * d["SQ_ACTIVE_INST_ANY"] / tot)
        cs = {k: int(v) for k, v in d.items() if k not in ("kernel", "grid", "us")}
        print("pass %s disp %4s %-52s grid %-8s %9.1f us %s %s" % (i, disp, d["kernel"], d["grid"], d["us"], cs, extra))


if __name__ == "__main__":
    main()
