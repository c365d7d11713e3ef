"""Layer-chain probe (1x MI355X): the eight forward launches of the NACF decoder layer at the bench's shapes (B = 128 videos,
2 x 128 sequences of 20 slots, ~50 % live, 120 memory rows per video) -- call by call on the default kernels, call by call
on the panel kernel, and as ONE chain launch under each barrier-fence mode (NACF_CHAIN_FENCE, csrc/gemm_bf16_chain.hpp), with
the chain's per-stage wall-clock stamps (NACF_CHAIN_TRACE=1).   usage: python tools/chain_probe.py [iters] [B]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nacf_amd  # noqa: E402,F401
from nacf_amd.runtime import lib as L, ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
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
tok = (torch.arange(Lq).unsqueeze(0) < lens.unsqueeze(1)).long() * 7
tok = tok.to(dev)
rows = ops.rowset_build(tokens=tok.reshape(-1))
n_live = int(tok.ne(0).sum())
x = (torch.rand(R * Lq, D, generator=g) - 0.5).to(dev)
kv = (torch.rand(B * Mk, 2 * D, generator=g) - 0.5).to(dev)
rng = ops.RngState(5, dev)
buf = {k: torch.empty(R * Lq, n, device=dev) for k, n in dict(qkv=3 * D, att=D, a=D, q=D, catt=D, c=D, u=FF, y=D).items()}
tf = tok.reshape(-1)
gelu = L.ACT_BY_NAME["gelu_new"]


def layer():
    ops.linear_fwd(x, W[0], buf["qkv"], ops.Epi(bias=bias[0]), rows, zero_dead=True)
    q = buf["qkv"]
    ops.attention_fwd(q[:, :D], q[:, D:2 * D], q[:, 2 * D:], buf["att"], tok, 0, None, R, H, Lq, Lq, D // H, 1, R)
    ops.linear_fwd(buf["att"], W[1], buf["a"], ops.Epi(bias=bias[1], p1=0.5, salt1=1, residual=x, row_tokens=tf, rng=rng), rows, zero_dead=True)
    ops.linear_fwd(buf["a"], W[2], buf["q"], ops.Epi(bias=bias[2]), rows, zero_dead=True)
    ops.attention_fwd(buf["q"], kv[:, :D], kv[:, D:], buf["catt"], None, 0, None, R, H, Lq, Mk, D // H, 1, B)
    ops.linear_fwd(buf["catt"], W[3], buf["c"], ops.Epi(bias=bias[3], p1=0.5, salt1=2, residual=buf["a"], row_tokens=tf, rng=rng), rows, zero_dead=True)
    ops.linear_fwd(buf["c"], W[4], buf["u"], ops.Epi(bias=bias[4], act=gelu), rows, zero_dead=True)
    ops.linear_fwd(buf["u"], W[5], buf["y"], ops.Epi(bias=bias[5], p1=0.5, salt1=3, residual=buf["c"], p2=0.5, salt2=4, row_tokens=tf, rng=rng), rows, zero_dead=True)


def timed(env, chained):
    for k in ("NACF_GEMM_PANEL", "NACF_CHAIN", "NACF_CHAIN_FENCE", "NACF_CHAIN_TRACE"):
        os.environ.pop(k, None)
    os.environ.update(env)

    def once():
        if chained:
            with ops.chain():
                layer()
        else:
            layer()
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    # one hipGraph of `iters` layers: what the step's replay sees (no host launch gaps)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(iters):
            once()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters, buf["y"].clone()


flops = 2.0 * n_live * sum(N * K for N, K in mats)
print("decoder layer forward, %d videos: %d of %d slots live, %.2f GF of nn.Linear work" % (B, n_live, R * Lq, flops / 1e9))
t0, y0 = timed(dict(NACF_GEMM_PANEL="0", NACF_CHAIN="0"), False)
print("  call by call, default kernels:  %7.1f us  (%5.1f TF over the whole layer incl. attention time)" % (t0, flops / t0 / 1e6))
t1, y1 = timed(dict(NACF_GEMM_PANEL="1", NACF_CHAIN="0"), False)
print("  call by call, panel kernel:     %7.1f us  (%5.1f TF)   max |dy| vs default %.2e" % (t1, flops / t1 / 1e6, float((y1 - y0).abs().max())))
for fence in ("0", "1", "2"):
    t2, y2 = timed(dict(NACF_GEMM_PANEL="1", NACF_CHAIN="1", NACF_CHAIN_FENCE=fence, NACF_CHAIN_TRACE="1"), True)
    st = (ctypes.c_uint64 * 25)()
    L.check(L.load().nacf_chain_stamps(st, 25, None), "nacf_chain_stamps")
    st = [int(v) for v in st]
    print("  ONE chain launch, fence mode %s: %7.1f us  (%5.1f TF)   bit-identical to the panel calls: %s   status %d"
          % (fence, t2, flops / t2 / 1e6, bool(torch.equal(y2, y1)), ops.chain_status()))
    names = ["q|k|v", "self-attention", "out-proj", "cross-q", "cross-attention", "out-proj", "FFN1", "FFN2"]
    for s_, nm in enumerate(names):
        a, b_, c = st[3 * s_], st[3 * s_ + 1], st[3 * s_ + 2]
        nxt = st[3 * (s_ + 1)]
        print("      stage %d %-16s workgroup 0 busy %6.2f us, last workgroup done after %6.2f us, next stage starts after %6.2f us"
              % (s_, nm, (b_ - a) / 100.0, (c - a) / 100.0, (nxt - a) / 100.0))
imgs.close()
