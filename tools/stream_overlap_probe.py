"""Does running a layer's dX and dW GEMMs on two HIP streams beat running them back to back? (tuning aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nacf_amd  # noqa: F401
from nacf_amd.runtime import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
r = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).to(dev)


def bench(f, iters=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


side = torch.cuda.Stream()
for (M, N, K) in [(2980, 2048, 512), (2980, 512, 2048), (2980, 1536, 512), (2980, 512, 512), (7680, 1024, 512)]:
    dz, w, x = r(M, N), r(N, K), r(M, K)
    dx, dw = torch.empty(M, K, device=dev), torch.empty(N, K, device=dev)

    def seq():
        ops.linear_bwd_data(dz, w, dx)
        ops.linear_bwd_weight(dz, x, dw, None, beta=0.0)

    def par():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            ops.linear_bwd_weight(dz, x, dw, None, beta=0.0)
        ops.linear_bwd_data(dz, w, dx)
        cur.wait_stream(side)

    t_dx = bench(lambda: ops.linear_bwd_data(dz, w, dx))
    t_dw = bench(lambda: ops.linear_bwd_weight(dz, x, dw, None, beta=0.0))
    t_seq, t_par = bench(seq), bench(par)
    gs, gp = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(gs):
        for _ in range(10):
            seq()
    with torch.cuda.graph(gp):
        for _ in range(10):
            par()
    t_gs, t_gp = bench(gs.replay, 10) / 10, bench(gp.replay, 10) / 10
    print("M,N,K=%s  dX %.1f us  dW %.1f us  seq %.1f  two-stream %.1f | graph: seq %.1f  two-stream %.1f"
          % ((M, N, K), t_dx, t_dw, t_seq, t_par, t_gs, t_gp))
