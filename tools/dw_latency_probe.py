import os, sys, torch
sys.path.insert(0, "/root/repo")
import nacf_amd
from nacf_amd.runtime import ops, lib as L
dev = torch.device("cuda", 0)
ops.set_gemm_mode("bf16x3")
M, N, K = 61440, 1024, 2048
dz, x, dw = torch.randn(M, N, device=dev), torch.randn(M, K, device=dev), torch.zeros(N, K, device=dev)
for label, rows in (("all rows distinct (HBM)", None),
                    ("row list 0..M-1 (HBM, through kmap)", ops.RowSet(torch.arange(M, dtype=torch.int32, device=dev), torch.tensor([M], dtype=torch.int32, device=dev))),
                    ("row list i % 64 (every load hits cache)", ops.RowSet((torch.arange(M, device=dev) % 64).int(), torch.tensor([M], dtype=torch.int32, device=dev)))):
    for wide in ("0", "1"):
        os.environ["NACF_DW_WIDE"] = wide
        ts = []
        for r in range(8):
            with ops.dw_group():
                ops.linear_bwd_weight(dz, x, dw, None, beta=0.0, rows=rows)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); L.load().nacf_dw_group_launch_gemms(ops._stream()); b.record()
            torch.cuda.synchronize()
            if r >= 2: ts.append(a.elapsed_time(b))
        ts.sort(); med = ts[len(ts)//2]
        # has_rs estimate makes the host assume 58 % live rows: only the split count changes
        print("%-42s wide=%s  %.3f ms  %.1f TF" % (label, wide, med, 2.0*M*N*K/med/1e9))
