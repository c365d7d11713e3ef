"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/rNN_pmc_traffic.json.

usage: python tools/pmc_traffic.py gpurun_out/pmc_fetch/b_counter_collection.csv \
                                   gpurun_out/pmc_write/b_counter_collection.csv profiles/r01_pmc_traffic.json [train|decode]
Both counters are reported in KiB by rocprofv3.  On gfx950 FETCH_SIZE counts exactly half of the bytes of a
wide coalesced read stream (MI355X_MICROARCH.md "HBM"): `hbm_bytes` below applies that x2 correction to the
read side and takes WRITE_SIZE as is (calibration on adam_step_kernel, whose traffic is known exactly --
4 reads + 3 writes of the 73.8 MB flat buffers -- gives 295 MB / 221 MB = the algorithmic figures)."""
import collections
import csv
import json
import sys
import time


def load(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def main():
    fetch, nf = load(sys.argv[1])
    write, _ = load(sys.argv[2])
    out = []
    for k in sorted(fetch, key=lambda k: -fetch[k] * nf[k]):
        f, w = fetch[k], write.get(k, 0.0)
        out.append({"kernel": k[0], "grid_threads": k[1], "launches_sampled": nf[k],
                    "fetch_size_kib": round(f, 1), "write_size_kib": round(w, 1),
                    "hbm_bytes": int((2.0 * f + w) * 1024)})
    leg = sys.argv[4] if len(sys.argv) > 4 else "train"
    cmd = {"train": "--steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-compare --no-loader --graph off",
           "decode": "--decode-only --decode-batches 3 --graph off"}[leg]
    json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py " + cmd,
               "leg": leg, "collected": time.strftime("%Y-%m-%d %H:%M UTC", time.gmtime()),
               "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024", "kernels": out}, open(sys.argv[3], "w"), indent=1)
    print("wrote", sys.argv[3], len(out), "kernel/grid classes")


if __name__ == "__main__":
    main()
