"""dW / dX / fwd GEMMs through a live-row list: tile x split sweep (tuning aid, GPU box only).
usage: python tools/dw_rows_bench.py kind:slots:live:N:K[,...]   (kind 0 fwd, 1 dX, 2 dW)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import nacf_amd  # noqa: E402,F401
from nacf_amd.runtime import ops  # noqa: E402

dev = torch.device("cuda:0")


def bench(kind, slots, live, N, K, iters=30):
    g = torch.Generator().manual_seed(0)
    r = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).to(dev)
    tok = torch.zeros(slots, dtype=torch.int64)
    tok[torch.randperm(slots, generator=g)[:live]] = 7
    rows = ops.rowset_build(tokens=tok.to(dev))
    if kind == 2:
        dz, x, dw, db = r(slots, ops.vocab_ld(N))[:, :N], r(slots, K), torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
        f = lambda: ops.linear_bwd_weight(dz, x, dw, db, beta=1.0, rows=rows)
    elif kind == 1:
        dz, w, dx = r(slots, ops.vocab_ld(N))[:, :N], r(N, K), torch.empty(slots, K, device=dev)
        f = lambda: ops.linear_bwd_data(dz, w, dx, rows=rows, zero_dead=True)
    else:
        x, w, y = r(slots, K), r(N, K), torch.empty(slots, ops.vocab_ld(N), device=dev)[:, :N]
        f = lambda: ops.linear_fwd(x, w, y, None, rows, zero_dead=True)
    for _ in range(3):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        f()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    return ms, 2.0 * live * N * K / (ms * 1e-3) / 1e12


for spec in sys.argv[1].split(","):
    kind, slots, live, N, K = (int(v) for v in spec.split(":"))
    for tile in os.environ.get("TILES", "64,128").split(","):
        for s in os.environ.get("SPLITS", "0,1,2,4,8").split(","):
            os.environ["NACF_GEMM_TILE"] = tile
            if s == "0":
                os.environ.pop("NACF_GEMM_SPLITS", None)
            else:
                os.environ["NACF_GEMM_SPLITS"] = s
            ms, tf = bench(kind, slots, live, N, K)
            print("%s tile %s splits %s: %.3f ms %.1f TF" % (spec, tile, s if s != "0" else "auto", ms, tf), flush=True)
