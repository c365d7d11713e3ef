"""Layer-chain probe 2: every stage of the decoder layer ALONE (a hipGraph of `iters` launches of that one call) -- default kernel,
panel kernel, and as a ONE-stage chain launch (same body inside the chain kernel) -- to separate what the chain kernel costs a
stage from what the barriers cost.   usage: python tools/chain_probe2.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nacf_amd  # noqa: E402,F401
from nacf_amd.runtime import lib as L, ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 128
dev = torch.device("cuda:0")
ops.set_gemm_mode("bf16x3")
D, H, Lq, Mk, FF = 512, 8, 20, 120, 2048
R = 2 * B
g = torch.Generator().manual_seed(1)
mats = [(3 * D, D), (D, D), (D, D), (D, D), (FF, D), (D, FF)]
offs, off = [], 0
for N, K in mats:
    offs.append(off)
    off += N * K
flat = ((torch.rand(off, generator=g) * 2 - 1) * 0.05).to(dev)
W = [flat[o:o + N * K].view(N, K) for o, (N, K) in zip(offs, mats)]
imgs = ops.WeightImages(flat, [(o, N, K, True) for o, (N, K) in zip(offs, mats)], 3)
imgs.refresh()
bias = [((torch.rand(N, generator=g) * 2 - 1) * 0.1).to(dev) for N, _ in mats]
lens = torch.randint(4, Lq, (R,), generator=g)
tok = ((torch.arange(Lq).unsqueeze(0) < lens.unsqueeze(1)).long() * 7).to(dev)
rows = ops.rowset_build(tokens=tok.reshape(-1))
x = (torch.rand(R * Lq, D, generator=g) - 0.5).to(dev)
kv = (torch.rand(B * Mk, 2 * D, generator=g) - 0.5).to(dev)
rng = ops.RngState(5, dev)
buf = {k: (torch.rand(R * Lq, n, generator=g) - 0.5).to(dev) for k, n in dict(qkv=3 * D, att=D, a=D, q=D, catt=D, c=D, u=FF, y=D).items()}
tf = tok.reshape(-1)
gelu = L.ACT_BY_NAME["gelu_new"]
q = buf["qkv"]
stages = [
    ("q|k|v", lambda: ops.linear_fwd(x, W[0], buf["qkv"], ops.Epi(bias=bias[0]), rows, zero_dead=True)),
    ("q|k|v, no dead-row fill", lambda: ops.linear_fwd(x, W[0], buf["qkv"], ops.Epi(bias=bias[0]), rows, zero_dead=False)),
    ("q|k|v, dense rows", lambda: ops.linear_fwd(x[:2944], W[0], buf["qkv"][:2944], ops.Epi(bias=bias[0]))),
    ("self-attention", lambda: ops.attention_fwd(q[:, :D], q[:, D:2 * D], q[:, 2 * D:], buf["att"], tok, 0, None, R, H, Lq, Lq, D // H, 1, R)),
    ("out-proj (dropout, residual)", lambda: ops.linear_fwd(buf["att"], W[1], buf["a"], ops.Epi(bias=bias[1], p1=0.5, salt1=1, residual=x, row_tokens=tf, rng=rng), rows, zero_dead=True)),
    ("out-proj, plain epilogue", lambda: ops.linear_fwd(buf["att"], W[1], buf["a"], ops.Epi(bias=bias[1]), rows, zero_dead=True)),
    ("cross-q", lambda: ops.linear_fwd(buf["a"], W[2], buf["q"], ops.Epi(bias=bias[2]), rows, zero_dead=True)),
    ("cross-attention", lambda: ops.attention_fwd(buf["q"], kv[:, :D], kv[:, D:], buf["catt"], None, 0, None, R, H, Lq, Mk, D // H, 1, B)),
    ("FFN1 (gelu)", lambda: ops.linear_fwd(buf["c"], W[4], buf["u"], ops.Epi(bias=bias[4], act=gelu), rows, zero_dead=True)),
    ("FFN2 (2 dropouts, residual)", lambda: ops.linear_fwd(buf["u"], W[5], buf["y"], ops.Epi(bias=bias[5], p1=0.5, salt1=3, residual=buf["c"], p2=0.5, salt2=4, row_tokens=tf, rng=rng), rows, zero_dead=True)),
]


def timed(env, chained, fn):
    for k in ("NACF_GEMM_PANEL", "NACF_CHAIN", "NACF_CHAIN_FENCE", "NACF_CHAIN_TRACE"):
        os.environ.pop(k, None)
    os.environ.update(env)

    def once():
        if chained:
            with ops.chain():
                fn()
        else:
            fn()
    for _ in range(3):
        once()
    name = L.load().nacf_gemm_last_kernel().decode()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(iters):
            once()
    gr.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / iters)
    return sorted(ts)[2], name


print("%d live rows of %d; one stage at a time, %d launches per graph, median of 5 replays" % (int(tok.ne(0).sum()), R * Lq, iters))
for nm, fn in stages:
    t0, k0 = timed(dict(NACF_GEMM_PANEL="0", NACF_CHAIN="0"), False, fn)
    t1, k1 = timed(dict(NACF_GEMM_PANEL="1", NACF_CHAIN="0"), False, fn)
    t2, k2 = timed(dict(NACF_GEMM_PANEL="1", NACF_CHAIN="1"), True, fn)
    print("  %-30s default %6.1f us   panel %6.1f us   one-stage chain %6.1f us     [%s | %s | %s]" % (nm, t0, t1, t2, k0[:34], k1[:24], k2[:30]))
imgs.close()
