"""GPU: the DMA-fed two-per-CU exact-mode GEMM (csrc/gemm_dma128.hpp: 64 / 128 x 128 tiles, both operands global -> LDS by DMA, the
activation split on the fragments, v_mfma_f32_32x32x16_bf16) behind nacf_linear_fwd / nacf_linear_bwd_data / nacf_vocab_lse_fwd /
nacf_vocab_argmax (the nn.Linear forward and dX of models/bert.py:139-247, models/__init__.py:83, decoding/algorithms.py:143-167).

Bars: fp64 reference at the fp32 kernels' tolerance; BIT-IDENTICAL to the wide kernel (same instruction, same per-accumulator
order of the (k-tile, k-step, term) products) wherever both are eligible; against the 128 x 128 / 64 x 64 kernels only rounding
differs; dropout masks are functions of the element index, so they are identical to the other kernels'; ragged reduce extents
(the vocabulary's dX reduces over 10547 = 329 x 32 + 19) are zero-filled in the request / cleaned in LDS.
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


def err(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


def tol(K, scale=1.0):
    return (1e-5 * math.sqrt(K) + 1e-5) * scale


def last_kernel(L):
    return L.load().nacf_gemm_last_kernel().decode()


class Weights:
    """a flat buffer of weight matrices with registered three-plane images (forward and transposed)"""

    def __init__(self, ops, dev, mats, seed=11, scale=0.5):
        offs, off = [], 0
        for N, K in mats:
            offs.append(off)
            off += (N * K + 3) // 4 * 4
        self.flat = rnd(off, seed=seed, scale=scale).to(dev)
        self.w = [self.flat[o:o + N * K].view(N, K) for o, (N, K) in zip(offs, mats)]
        self.imgs = ops.WeightImages(self.flat, [(o, N, K, True) for o, (N, K) in zip(offs, mats)], 3)
        self.imgs.refresh()

    def close(self):
        self.imgs.close()


# (M, N, K): whole tiles, ragged rows / columns, a reduce extent of one k-tile, odd k-tile counts, K % 32 != 0, K % 4 != 0 for dX
SHAPES = [(1000, 512, 256), (700, 300, 128), (333, 1030, 192), (600, 256, 2048), (129, 257, 320), (64, 128, 32), (257, 131, 96),
          (200, 203, 160)]


@pytest.mark.parametrize("mt", ["1", "2"])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_dma128_forward_and_dx_vs_fp64_wide_and_the_other_kernels(dev, M, N, K, mt, monkeypatch):
    ops, L = _ops()
    monkeypatch.setenv("NACF_GEMM_MODE", "bf16x3")
    W = Weights(ops, dev, [(N, K)])
    try:
        w = W.w[0]
        x, b = rnd(M, K, seed=1).to(dev), rnd(N, seed=3).to(dev)
        ldz = N + (-N) % 4
        dzbuf = torch.full((M + 1, ldz), float("nan"), device=dev)        # whatever follows a row in memory must not matter
        dzbuf[:M, :N] = rnd(M, N, seed=4).to(dev)
        dz = dzbuf[:M, :N]
        ref_y = x.double().cpu() @ w.double().cpu().t() + b.double().cpu()
        ref_dx = dz.double().cpu() @ w.double().cpu()
        out = {}
        for name, env in (("dma", {"NACF_DMA128": mt, "NACF_GEMM_WIDE": "0"}), ("wide", {"NACF_DMA128": "0", "NACF_GEMM_WIDE": mt}),
                          ("old", {"NACF_DMA128": "0", "NACF_GEMM_WIDE": "0"})):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            y, dx = torch.full((M, N), 7.0, device=dev), torch.full((M, K), 7.0, device=dev)
            ops.linear_fwd(x, w, y, ops.Epi(bias=b))
            k1 = last_kernel(L)
            ops.linear_bwd_data(dz, w, dx)
            k2 = last_kernel(L)
            out[name] = (y, dx, k1, k2)
        y, dx, k1, k2 = out["dma"]
        assert k1 == "gemm_dma128_kernel<%s, EpiLinear>" % mt, k1
        assert k2 == "gemm_dma128_kernel<%s, EpiStore>" % mt, k2
        assert out["old"][2].startswith("gemm_bf16_kernel") and out["old"][3].startswith("gemm_bf16_kernel")
        assert err(y, ref_y) < tol(K) and err(dx, ref_dx) < tol(N)
        assert err(y, out["old"][0]) < tol(K, 0.2) and err(dx, out["old"][1]) < tol(N, 0.2)
        if out["wide"][2].startswith("gemm_wide"):
            assert torch.equal(y, out["wide"][0]), "forward differs from the wide kernel"
        if out["wide"][3].startswith("gemm_wide"):
            assert torch.equal(dx, out["wide"][1]), "dX differs from the wide kernel"
    finally:
        W.close()


def test_dma128_is_not_taken_when_not_eligible(dev, monkeypatch):
    """no registered image, the other arithmetic modes, NACF_GEMM_TILE set: the other kernels run"""
    ops, L = _ops()
    monkeypatch.setenv("NACF_DMA128", "2")
    monkeypatch.setenv("NACF_GEMM_WIDE", "0")
    W = Weights(ops, dev, [(256, 256)])
    try:
        x = rnd(300, 256, seed=2).to(dev)
        y = torch.empty(300, 256, device=dev)
        for mode in ("bf16", "f32"):
            monkeypatch.setenv("NACF_GEMM_MODE", mode)
            ops.linear_fwd(x, W.w[0], y, None)
            assert not last_kernel(L).startswith("gemm_dma128"), (mode, last_kernel(L))
        monkeypatch.setenv("NACF_GEMM_MODE", "bf16x3")
        ops.linear_fwd(x, W.w[0], y, None)
        assert last_kernel(L).startswith("gemm_dma128")
        monkeypatch.setenv("NACF_GEMM_TILE", "64")
        ops.linear_fwd(x, W.w[0], y, None)
        assert last_kernel(L).startswith("gemm_bf16_kernel")
        monkeypatch.delenv("NACF_GEMM_TILE")
        w = rnd(256, 256, seed=9).to(dev)           # not registered: converted inside the (other) kernel
        ops.linear_fwd(x, w, y, None)
        assert last_kernel(L).startswith("gemm_bf16_kernel")
        assert err(y, x.double().cpu() @ w.double().cpu().t()) < tol(256)
    finally:
        W.close()


@pytest.mark.parametrize("mt", ["1", "2"])
def test_dma128_fused_epilogue_rows_and_dropout(dev, mt, monkeypatch):
    """every field of the nn.Linear epilogue, a live-row list with dead-row fill (dead rows are never read), accumulate, and
    dropout masks equal to the other kernels'"""
    ops, L = _ops()
    monkeypatch.setenv("NACF_GEMM_MODE", "bf16x3")
    monkeypatch.setenv("NACF_GEMM_WIDE", "0")
    M, N, K = 900, 512, 256
    W = Weights(ops, dev, [(N, K)], scale=0.3)
    try:
        w = W.w[0]
        x, b, r = rnd(M, K, seed=4), rnd(N, seed=6), rnd(M, N, seed=7)
        tok = torch.randint(0, 3, (M,), generator=torch.Generator().manual_seed(5))
        z = x.double() @ w.double().cpu().t() + b.double()
        gelu = 0.5 * z * (1 + torch.tanh(math.sqrt(2 / math.pi) * (z + 0.044715 * z ** 3)))
        ref = (gelu + r.double()) * tok.ne(0).double().unsqueeze(1)
        monkeypatch.setenv("NACF_DMA128", mt)
        y, pre = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
        ops.linear_fwd(x.to(dev), w, y, ops.Epi(bias=b.to(dev), act=L.ACT_GELU_NEW, preact=pre, residual=r.to(dev),
                                                row_tokens=tok.to(dev)))
        assert last_kernel(L) == "gemm_dma128_kernel<%s, EpiLinear>" % mt
        assert err(pre, z) < tol(K) and err(y, ref) < tol(K)
        live = tok.ne(PAD).nonzero().squeeze(1)
        xn = x.clone()
        xn[tok.eq(PAD)] = float("nan")
        rows = ops.rowset_build(tokens=tok.to(dev))
        y = torch.full((M, N), 7.0, device=dev)
        ops.linear_fwd(xn.to(dev), w, y, ops.Epi(bias=b.to(dev)), rows, zero_dead=True)
        assert last_kernel(L).startswith("gemm_dma128_kernel<%s" % mt)
        assert err(y[live.to(dev)], x[live].double() @ w.double().cpu().t() + b.double()) < tol(K)
        assert float(y[tok.eq(PAD).to(dev)].abs().max()) == 0.0
        dz = rnd(M, N, seed=8)
        dz[tok.eq(PAD)] = float("nan")
        dx = torch.full((M, K), 7.0, device=dev)
        ops.linear_bwd_data(dz.to(dev), w, dx, rows=rows, zero_dead=True)
        assert last_kernel(L) == "gemm_dma128_kernel<%s, EpiStore>" % mt
        assert err(dx[live.to(dev)], dz[live].double() @ w.double().cpu()) < tol(N)
        assert float(dx[tok.eq(PAD).to(dev)].abs().max()) == 0.0
        base = rnd(M, K, seed=9).to(dev)
        dz2 = rnd(M, N, seed=10).to(dev)
        acc = base.clone()
        ops.linear_bwd_data(dz2, w, acc, beta=1.0)
        assert err(acc, base.double().cpu() + dz2.double().cpu() @ w.double().cpu()) < tol(N)
        rng = ops.RngState(1234, dev)
        outs = {}
        for d in (mt, "0"):
            monkeypatch.setenv("NACF_DMA128", d)
            yd = torch.empty(M, N, device=dev)
            ops.linear_fwd(x.to(dev), w, yd, ops.Epi(bias=b.to(dev), p1=0.5, salt1=11, residual=r.to(dev), p2=0.25, salt2=12, rng=rng))
            outs[d] = yd
        assert torch.equal(outs[mt].eq(0), outs["0"].eq(0))
        assert err(outs[mt], outs["0"]) < tol(K, 2.0) and bool(outs[mt].ne(0).any())
    finally:
        W.close()


@pytest.mark.parametrize("mt", ["1", "2"])
def test_dma128_splitk_dx_with_a_ragged_reduce_extent(dev, mt, monkeypatch):
    """dX over a long reduce dimension that is no multiple of 4 (the vocabulary: 10547), with and without reduce splits"""
    ops, L = _ops()
    monkeypatch.setenv("NACF_GEMM_MODE", "bf16x3")
    monkeypatch.setenv("NACF_GEMM_WIDE", "0")
    monkeypatch.setenv("NACF_DMA128", mt)
    M, N, K = 300, 4099, 256
    W = Weights(ops, dev, [(N, K)], scale=0.1)
    try:
        ldz = N + (-N) % 4
        dzbuf = torch.full((M, ldz), float("nan"), device=dev)
        dzbuf[:, :N] = rnd(M, N, seed=3).to(dev)
        dz = dzbuf[:, :N]
        ref = dz.double().cpu() @ W.w[0].double().cpu()
        for splits in ("1", "4"):
            monkeypatch.setenv("NACF_GEMM_SPLITS", splits)
            dx = torch.empty(M, K, device=dev)
            ops.linear_bwd_data(dz, W.w[0], dx)
            assert last_kernel(L) == "gemm_dma128_kernel<%s, EpiStore>" % mt, last_kernel(L)
            assert err(dx, ref) < tol(N, 0.2), splits
    finally:
        W.close()


@pytest.mark.parametrize("mt", ["1", "2"])
def test_dma128_vocabulary_projection_with_softmax_statistics(dev, mt, monkeypatch):
    """nacf_vocab_lse_fwd / nacf_vocab_argmax on the kernel's EpiArgmax epilogue: logits, log-sum-exp, label log-probability and arg-max
    against fp64, and the same arg-max as the 128 x 128 kernel"""
    ops, L = _ops()
    monkeypatch.setenv("NACF_GEMM_MODE", "bf16x3")
    monkeypatch.setenv("NACF_GEMM_WIDE", "0")
    rows_n, V, K = 600, 1003, 128
    W = Weights(ops, dev, [(V, K)], scale=0.5)
    try:
        g = torch.Generator().manual_seed(3)
        h, bias = rnd(rows_n, K, seed=5).to(dev), rnd(V, seed=6).to(dev)
        lab = torch.randint(1, V, (rows_n,), generator=g).to(dev)
        tok = (torch.rand(rows_n, generator=g) < 0.6).long().to(dev)
        rows = ops.rowset_build(tokens=tok)
        ref = h.double().cpu() @ W.w[0].double().cpu().t() + bias.double().cpu()
        ref_lse = torch.logsumexp(ref, 1)
        res = {}
        for d in (mt, "0"):
            monkeypatch.setenv("NACF_DMA128", d)
            buf = torch.zeros(rows_n, ops.vocab_ld(V), device=dev)
            lse, llp = torch.zeros(rows_n, device=dev), torch.zeros(rows_n, device=dev)
            am = torch.zeros(rows_n, dtype=torch.int64, device=dev)
            ops.vocab_lse_fwd(h, W.w[0], bias, buf[:, :V], lab, lse, am, llp, rows)
            name = last_kernel(L)
            res[d] = (buf[:, :V].clone(), lse, llp, am, name)
        logits, lse, llp, am, name = res[mt]
        assert name == "gemm_dma128_kernel<%s, EpiArgmax>" % mt, name
        assert res["0"][4].startswith("gemm_bf16_kernel")
        live = tok.ne(0).cpu()
        assert err(logits[live.to(dev)], ref[live]) < tol(K)
        assert err(lse[live.to(dev)], ref_lse[live]) < 1e-5
        assert err(llp[live.to(dev)], (ref - ref_lse[:, None]).gather(1, lab.cpu()[:, None]).squeeze(1)[live]) < 2e-5
        assert torch.equal(am.cpu()[live], ref.argmax(1)[live])
        assert torch.equal(am, res["0"][3])
    finally:
        W.close()
