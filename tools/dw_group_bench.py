"""Times the grouped weight-gradient launch (nacf_dw_group_launch_gemms alone, HIP events): one long problem (the k-loop's
rate), the step's problems one at a time, and the NACF step's set of problems.
usage (GPU box): python tools/dw_group_bench.py [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nacf_amd  # noqa: E402,F401
from nacf_amd.runtime import ops, lib as L  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
ops.set_gemm_mode("bf16x3")
SETS = {
    "long k-loop (61440 x 1024 x 2048)": [(61440, 1024, 2048)],
    "enc_lin (7680 x 512 x 2048)": [(7680, 512, 2048)],
    "hw (7680 x 1024 x 512)": [(7680, 1024, 512)],
    "kvmem (15360 x 1024 x 512)": [(15360, 1024, 512)],
    "qkv (5120 x 1536 x 512)": [(5120, 1536, 512)],
    "proj (5120 x 512 x 512)": [(5120, 512, 512)],
    "ffn1 (5120 x 2048 x 512)": [(5120, 2048, 512)],
    "ffn2 (5120 x 512 x 2048)": [(5120, 512, 2048)],
    "NACF step": [(7680, 512, 2048)] * 2 + [(7680, 1024, 512)] * 2 + [(15360, 1024, 512), (5120, 1536, 512), (5120, 512, 512),
                  (5120, 512, 512), (5120, 512, 512), (5120, 2048, 512), (5120, 512, 2048), (128, 512, 512)],      # (without the vocabulary projection: it needs a padded pitch and its row list)
}
for title, probs in SETS.items():
    ts = [(torch.randn(M, N, device=dev), torch.randn(M, K, device=dev), torch.zeros(N, K, device=dev)) for M, N, K in probs]
    flops = sum(2.0 * M * N * K for M, N, K in probs)
    for wide, wgs in (("-", None),):
        if wgs: os.environ["NACF_DW_GROUP_WGS"] = wgs
        else: os.environ.pop("NACF_DW_GROUP_WGS", None)
        best = []
        for r in range(reps + 3):
            with ops.dw_group():
                for dz, x, dw in ts:
                    ops.linear_bwd_weight(dz, x, dw, None, beta=0.0)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                L.load().nacf_dw_group_launch_gemms(ops._stream())
                b.record()
            torch.cuda.synchronize()
            if r >= 3: best.append(a.elapsed_time(b))
        best.sort()
        med = best[len(best) // 2]
        print("%-36s wide=%s wgs=%-5s median %.3f ms  %.1f TF  (min %.3f)" % (title, wide, wgs or "dflt", med, flops / med / 1e9, best[0]))
