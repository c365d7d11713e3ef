"""Throughput (bf16) mode, the weight gradients of one NACF backward pass as ONE dw_group (grouped GEMM + combine), the
g256 path (NACF_DW_G256=3: csrc/nacf_gemm_g256.hip, gemm_g256w.hpp) against round 4's grouped 128 x 128 kernel (=0), interleaved rounds.
usage (GPU box): python tools/dw_g256_bench.py [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nacf_amd  # noqa: E402,F401
from nacf_amd.runtime import ops, lib as L  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
ops.set_gemm_mode(os.environ.get("MODE", "bf16"))
g = torch.Generator().manual_seed(3)
# (rows, N, K, live-row list?)  NACF, 128 videos: decoder rows 2 x 128 x 20 = 5120 slots (~58 % live), 7680 frames per modality
STEP = [(7680, 512, 2048, 0)] * 2 + [(7680, 1024, 512, 0)] * 2 + [(15360, 1024, 512, 0), (5120, 1536, 512, 1), (5120, 512, 512, 1),
        (5120, 512, 512, 1), (5120, 512, 512, 1), (5120, 2048, 512, 1), (5120, 512, 2048, 1), (5120, 10547, 512, 1)]
if os.environ.get("ONLY"):
    STEP = [STEP[int(i)] for i in os.environ["ONLY"].split(",")]
NOBIAS, NOROWS = os.environ.get("NOBIAS") == "1", os.environ.get("NOROWS") == "1"
tok = (torch.rand(5120, generator=g) < 0.58).to(dev).long()
rows = ops.rowset_build(tokens=tok)
ts = []
for M, N, K, rl in STEP:
    ts.append((torch.randn(M, ops.vocab_ld(N), device=dev)[:, :N], torch.randn(M, K, device=dev), torch.zeros(N, K, device=dev),
               None if NOBIAS else torch.zeros(N, device=dev), rows if (rl and not NOROWS) else None))
live = int(rows.count)
flops = sum(2.0 * (live if (rl and not NOROWS) else M) * N * K for M, N, K, rl in STEP)
print("NACF step, 128 videos: %d problems, %d of 5120 decoder slots live, %.1f GF of live-row work" % (len(STEP), live, flops / 1e9))


def issue(flag):
    os.environ["NACF_DW_G256"] = flag
    with ops.dw_group():
        for dz, x, dw, db, rs in ts:
            ops.linear_bwd_weight(dz, x, dw, db, beta=1.0, rows=rs)


graphs = {}
for flag in ("3", "0"):
    for _ in range(3):                       # sizes the group's buffers outside the capture
        issue(flag)
    torch.cuda.synchronize()
    print("flag", flag, "last kernel:", (L.load().nacf_gemm_last_kernel() or b"").decode())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        issue(flag)
    graphs[flag] = gr
res = {"3": [], "0": []}
for r in range(reps + 3):
    for flag in ("3", "0"):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        graphs[flag].replay()
        b.record()
        torch.cuda.synchronize()
        if r >= 3:
            res[flag].append(a.elapsed_time(b))
for flag, name in (("3", "256x256 eight-phase on fp32 operands + combine"), ("0", "round-4 grouped 128x128 kernel + combine")):
    v = sorted(res[flag])
    med = v[len(v) // 2]
    print("%-52s median %.3f ms  min %.3f ms  %.1f TF of live-row work (graph replay)" % (name, med, v[0], flops / med / 1e9))
