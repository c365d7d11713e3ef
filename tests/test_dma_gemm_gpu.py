"""GPU: the DMA-staged throughput-mode GEMM (csrc/gemm_bf16_dma.hpp: both operands global -> LDS by DMA, fp32 activations
rounded to bf16 on the fragments, three LDS images, v_mfma_f32_32x32x16_bf16) behind nacf_linear_fwd / nacf_linear_bwd_data
in the `bf16` arithmetic mode.

Bars: the kernel feeds the matrix cores the SAME bf16 values as the register-staged kernels of the mode (RNE of the fp32
operands, fp32 accumulate), so against an fp64 product of the rounded operands it holds the fp32 kernels' tolerance, and
against the register-staged kernels only the summation order differs; dropout masks are functions of the element index, so
they are identical to the other kernels'.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
PAD = 0


def _ops():
    import nacf_amd  # noqa: F401
    from nacf_amd.runtime import ops, lib
    return ops, lib


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def bf(x):
    """the value the throughput mode multiplies: round-to-nearest-even bf16 of the fp32 operand, as float64"""
    return x.detach().float().cpu().to(torch.bfloat16).double()


def err(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


def tol(K, scale=1.0):
    return (1e-5 * math.sqrt(K) + 1e-5) * scale


def last_kernel(L):
    return L.load().nacf_gemm_last_kernel().decode()


class Weights:
    """a flat buffer of weight matrices with registered one-plane images (the DMA kernel reads images only)"""

    def __init__(self, ops, dev, mats, seed=11, scale=0.5):
        offs, off = [], 0
        for N, K in mats:
            offs.append(off)
            off += (N * K + 3) // 4 * 4
        self.flat = rnd(off, seed=seed, scale=scale).to(dev)
        self.w = [self.flat[o:o + N * K].view(N, K) for o, (N, K) in zip(offs, mats)]
        self.imgs = ops.WeightImages(self.flat, [(o, N, K, True) for o, (N, K) in zip(offs, mats)], 1)
        self.imgs.refresh()

    def close(self):
        self.imgs.close()


SHAPES = [(1000, 512, 256), (700, 300, 128), (333, 1030, 192), (600, 256, 2048), (129, 257, 320), (64, 256, 64), (257, 96, 32), (5120, 512, 512)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_dma_forward_and_dx_vs_fp64_and_the_other_kernels(dev, M, N, K, monkeypatch):
    ops, L = _ops()
    monkeypatch.setenv("NACF_GEMM_MODE", "bf16")
    W = Weights(ops, dev, [(N, K)])
    try:
        w = W.w[0]
        x, b = rnd(M, K, seed=1).to(dev), rnd(N, seed=3).to(dev)
        dz = rnd(M, N + (-N) % 4, seed=4).to(dev)[:, :N]
        ref_y = bf(x) @ bf(w).t() + b.double().cpu()
        ref_dx = bf(dz) @ bf(w)
        out = {}
        for dma in ("1", "0"):
            monkeypatch.setenv("NACF_GEMM_DMA", dma)
            y, dx = torch.full((M, N), 7.0, device=dev), torch.full((M, K), 7.0, device=dev)
            ops.linear_fwd(x, w, y, ops.Epi(bias=b))
            k1 = last_kernel(L)
            ops.linear_bwd_data(dz, w, dx)
            k2 = last_kernel(L)
            out[dma] = (y, dx, k1, k2)
        y, dx, k1, k2 = out["1"]
        assert k1.startswith("gemm_dma_kernel<EpiLinear"), k1
        if N % 32 == 0:          # dX reduces over N: whole 32-wide k-tiles
            assert k2.startswith("gemm_dma_kernel<EpiStore"), k2
        else:
            assert k2.startswith("gemm_bf16_kernel"), k2
        assert out["0"][2].startswith("gemm_bf16_kernel") and out["0"][3].startswith("gemm_bf16_kernel")
        assert err(y, ref_y) < tol(K) and err(dx, ref_dx) < tol(N)
        assert err(y, out["0"][0]) < tol(K, 0.2) and err(dx, out["0"][1]) < tol(N, 0.2)
    finally:
        W.close()


def test_dma_is_not_taken_when_not_eligible(dev, monkeypatch):
    """a reduce dimension that is not whole 32-wide k-tiles, no registered image, the other modes: the register-staged kernels"""
    ops, L = _ops()
    monkeypatch.setenv("NACF_GEMM_DMA", "1")
    W = Weights(ops, dev, [(256, 80), (256, 48), (256, 256)])
    try:
        for mode, w in [("bf16", W.w[0]), ("bf16", W.w[1]), ("f32", W.w[2])]:
            monkeypatch.setenv("NACF_GEMM_MODE", mode)
            N, K = w.shape
            x = rnd(300, K, seed=2).to(dev)
            y = torch.empty(300, N, device=dev)
            ops.linear_fwd(x, w, y, None)
            assert not last_kernel(L).startswith("gemm_dma"), (mode, K, last_kernel(L))
            assert err(y, bf(x) @ bf(w).t() if mode == "bf16" else x.double().cpu() @ w.double().cpu().t()) < tol(K)
        monkeypatch.setenv("NACF_GEMM_MODE", "bf16")
        w = rnd(256, 256, seed=9).to(dev)           # not registered: converted inside the (other) kernel
        x = rnd(300, 256, seed=2).to(dev)
        y = torch.empty(300, 256, device=dev)
        ops.linear_fwd(x, w, y, None)
        assert last_kernel(L).startswith("gemm_bf16_kernel")
    finally:
        W.close()


def test_dma_fused_epilogue_rows_and_dropout(dev, monkeypatch):
    """every field of the nn.Linear epilogue, a live-row list with dead-row fill, beta = 1, and dropout masks equal to the
    other kernels' (they depend on the element index only)"""
    ops, L = _ops()
    monkeypatch.setenv("NACF_GEMM_MODE", "bf16")
    monkeypatch.setenv("NACF_GEMM_DMA", "1")
    M, N, K = 900, 512, 256
    W = Weights(ops, dev, [(N, K)], scale=0.3)
    try:
        w = W.w[0]
        x, b, r = rnd(M, K, seed=4), rnd(N, seed=6), rnd(M, N, seed=7)
        tok = torch.randint(0, 3, (M,), generator=torch.Generator().manual_seed(5))
        z = bf(x) @ bf(w).t() + b.double()
        gelu = 0.5 * z * (1 + torch.tanh(math.sqrt(2 / math.pi) * (z + 0.044715 * z ** 3)))
        ref = (gelu + r.double()) * tok.ne(0).double().unsqueeze(1)
        y, pre = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
        ops.linear_fwd(x.to(dev), w, y, ops.Epi(bias=b.to(dev), act=L.ACT_GELU_NEW, preact=pre, residual=r.to(dev),
                                                row_tokens=tok.to(dev)))
        assert last_kernel(L).startswith("gemm_dma_kernel<EpiLinear")
        assert err(pre, z) < tol(K) and err(y, ref) < tol(K)
        # live rows: dead rows are never read and zero-filled on request
        live = tok.ne(PAD).nonzero().squeeze(1)
        xn = x.clone()
        xn[tok.eq(PAD)] = float("nan")
        rows = ops.rowset_build(tokens=tok.to(dev))
        y = torch.full((M, N), 7.0, device=dev)
        ops.linear_fwd(xn.to(dev), w, y, ops.Epi(bias=b.to(dev)), rows, zero_dead=True)
        assert last_kernel(L).startswith("gemm_dma_kernel<EpiLinear")
        assert err(y[live.to(dev)], bf(x[live]) @ bf(w).t() + b.double()) < tol(K)
        assert float(y[tok.eq(PAD).to(dev)].abs().max()) == 0.0
        dz = rnd(M, N, seed=8)
        dz[tok.eq(PAD)] = float("nan")
        dx = torch.full((M, K), 7.0, device=dev)
        ops.linear_bwd_data(dz.to(dev), w, dx, rows=rows, zero_dead=True)
        assert last_kernel(L).startswith("gemm_dma_kernel<EpiStore")
        assert err(dx[live.to(dev)], bf(dz[live]) @ bf(w)) < tol(N)
        assert float(dx[tok.eq(PAD).to(dev)].abs().max()) == 0.0
        # accumulate (beta = 1), as the HighWay backward does
        base = rnd(M, K, seed=9).to(dev)
        dz2 = rnd(M, N, seed=10).to(dev)
        acc = base.clone()
        ops.linear_bwd_data(dz2, w, acc, beta=1.0)
        assert err(acc, base.double().cpu() + bf(dz2) @ bf(w)) < tol(N)
        # dropout: same masks as the other kernels
        rng = ops.RngState(1234, dev)
        outs = {}
        for dma in ("1", "0"):
            monkeypatch.setenv("NACF_GEMM_DMA", dma)
            yd = torch.empty(M, N, device=dev)
            ops.linear_fwd(x.to(dev), w, yd, ops.Epi(bias=b.to(dev), p1=0.5, salt1=11, residual=r.to(dev), p2=0.25, salt2=12, rng=rng))
            outs[dma] = yd
        assert torch.equal(outs["1"].eq(0), outs["0"].eq(0))
        assert err(outs["1"], outs["0"]) < tol(K, 2.0) and bool(outs["1"].ne(0).any())
    finally:
        W.close()


def test_dma_splitk_dx(dev, monkeypatch):
    """dX over a long reduce dimension with reduce splits (the slab + combine path)"""
    ops, L = _ops()
    monkeypatch.setenv("NACF_GEMM_MODE", "bf16")
    monkeypatch.setenv("NACF_GEMM_DMA", "1")
    monkeypatch.setenv("NACF_GEMM_SPLITS", "4")
    M, N, K = 300, 4096, 256
    W = Weights(ops, dev, [(N, K)], scale=0.1)
    try:
        dz = rnd(M, N, seed=3).to(dev)
        dx = torch.empty(M, K, device=dev)
        ops.linear_bwd_data(dz, W.w[0], dx)
        assert last_kernel(L).startswith("gemm_dma_kernel<EpiStore"), last_kernel(L)
        assert err(dx, bf(dz) @ bf(W.w[0])) < tol(N, 0.2)
    finally:
        W.close()
