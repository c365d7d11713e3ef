"""GPU: the panel GEMM (csrc/gemm_bf16_panel.hpp: persistent, the four waves of a workgroup split the reduce dimension, weights
streamed from a fragment-major image into the accumulation file, operand stream running on across tile boundaries) behind
nacf_linear_fwd / nacf_linear_bwd_data -- the skinny launches of the decoder layer (models/bert.py:139-247).

Bars: fp64 reference at the fp32 kernels' tolerance (the kernel is a drop-in for the parity path); against the 64x64 /
128x128 kernels of the same mode only rounding differs (another matrix-instruction shape and summation tree); dropout masks
are functions of the element index, so they are IDENTICAL to the other kernels'; the fragment-major images the library
builds equal an independent construction bit for bit.
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
    def __init__(self, ops, dev, mats, seed=11, scale=0.5):
        offs, off = [], 0
        for N, K in mats:
            offs.append(off)
            off += (N * K + 3) // 4 * 4
        self.flat = rnd(off, seed=seed, scale=scale).to(dev)
        self.offs = offs
        self.w = [self.flat[o:o + N * K].view(N, K) for o, (N, K) in zip(offs, mats)]
        # fragment-major images of EVERY eligible matrix (by default only the reduce dimensions the heuristic takes get one)
        import os
        before = os.environ.get("NACF_GEMM_PANEL")
        os.environ["NACF_GEMM_PANEL"] = "1"
        try:
            self.imgs = ops.WeightImages(self.flat, [(o, N, K, True) for o, (N, K) in zip(offs, mats)], 3)
        finally:
            if before is None:
                del os.environ["NACF_GEMM_PANEL"]
            else:
                os.environ["NACF_GEMM_PANEL"] = before
        self.imgs.refresh()

    def close(self):
        self.imgs.close()


def split3(w):
    """the exact three-way bf16 split by truncation (csrc/gemm_bf16.hpp: bf16_split2), as int16 bit patterns [3, ...]"""
    x = w.detach().float().cpu().clone()
    planes = []
    for _ in range(3):
        bits = x.view(torch.int32) & -65536
        planes.append((bits >> 16).to(torch.int16))
        x = x - bits.view(torch.float32)
    return torch.stack(planes)


def frag_image(w):
    """[k / 16][n / 32][term][lane = ((k % 16) / 8) * 32 + n % 32][k % 8] of W [N, K] (include/nacf_hip.h)"""
    N, K = w.shape
    p = split3(w)                                             # [3, N, K]
    p = p.view(3, N // 32, 32, K // 16, 2, 8)                 # s, nt, l31, k16, h, e
    return p.permute(3, 1, 0, 4, 2, 5).contiguous().view(-1)


# (M, N, K): decoder-layer shapes and a few ragged row counts; K = 512 (the ring holds a tile), 768 (odd chunk count:
# the ring parity changes from tile to tile), 1024 / 2048 (refills); N up to 4 column blocks per panel
SHAPES = [(2560, 512, 512), (1000, 128, 512), (333, 256, 768), (700, 512, 2048), (64, 128, 512), (31, 384, 1024), (4100, 1536, 512)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_panel_forward_and_dx_vs_fp64_and_the_other_kernels(dev, M, N, K, monkeypatch):
    ops, L = _ops()
    monkeypatch.setenv("NACF_GEMM_MODE", "bf16x3")
    W = Weights(ops, dev, [(N, K)])
    try:
        w = W.w[0]
        x, b = rnd(M, K, seed=1).to(dev), rnd(N, seed=3).to(dev)
        dz = rnd(M, N, seed=4).to(dev)
        ref_y = x.double().cpu() @ w.double().cpu().t() + b.double().cpu()
        ref_dx = dz.double().cpu() @ w.double().cpu()
        out = {}
        for panel in ("1", "0"):
            monkeypatch.setenv("NACF_GEMM_PANEL", panel)
            y, dx = torch.full((M, N), 7.0, device=dev), torch.full((M, K), 7.0, device=dev)
            ops.linear_fwd(x, w, y, ops.Epi(bias=b))
            k1 = last_kernel(L)
            ops.linear_bwd_data(dz, w, dx)
            k2 = last_kernel(L)
            out[panel] = (y, dx, k1, k2)
        y, dx, k1, k2 = out["1"]
        assert k1.startswith("gemm_panel_kernel<2, 4, 3, EpiLinear"), k1
        # dX: P = W^T [K rows, reduce over N]: eligible when K % 128 == 0 and N % 256 == 0, N >= 512
        if K % 128 == 0 and N % 256 == 0 and N >= 512:
            assert k2.startswith("gemm_panel_kernel<2, 4, 3, EpiStore"), k2
        else:
            assert not k2.startswith("gemm_panel"), k2
        assert not out["0"][2].startswith("gemm_panel") and not out["0"][3].startswith("gemm_panel")
        assert err(y, ref_y) < tol(K) and err(dx, ref_dx) < tol(N)
        assert err(y, out["0"][0]) < tol(K, 0.2) and err(dx, out["0"][1]) < tol(N, 0.2)
    finally:
        W.close()


def test_fragment_major_images_equal_an_independent_construction(dev, monkeypatch):
    ops, L = _ops()
    monkeypatch.setenv("NACF_GEMM_MODE", "bf16x3")
    mats = [(512, 512), (1536, 512), (512, 2048), (300, 512), (128, 768)]
    W = Weights(ops, dev, mats)
    try:
        fo = to = 0
        for (N, K), w in zip(mats, W.w):
            if N % 128 == 0 and K % 256 == 0 and K >= 512:
                n = (K // 16) * (N // 32) * 3 * 512
                assert torch.equal(W.imgs.fimg[fo:fo + n].cpu(), frag_image(w)), (N, K)
                fo += n
            if K % 128 == 0 and N % 256 == 0 and N >= 512:
                n = (N // 16) * (K // 32) * 3 * 512
                assert torch.equal(W.imgs.fimgT[to:to + n].cpu(), frag_image(w.t().contiguous())), (N, K, "T")
                to += n
        assert fo > 0 and to > 0
    finally:
        W.close()


def test_panel_is_not_taken_when_not_eligible(dev, monkeypatch):
    """rows not a multiple of 128, a reduce dimension that is not a multiple of 256 or below 512, no registered image, the
    other modes, a slice of a registered matrix that does not start on a 32-row block: the other kernels run"""
    ops, L = _ops()
    monkeypatch.setenv("NACF_GEMM_PANEL", "1")
    W = Weights(ops, dev, [(192, 512), (256, 384), (256, 256), (256, 512)])
    try:
        for mode, w in [("bf16x3", W.w[0]), ("bf16x3", W.w[1]), ("bf16x3", W.w[2]), ("bf16", W.w[3]), ("f32", W.w[3])]:
            monkeypatch.setenv("NACF_GEMM_MODE", mode)
            N, K = w.shape
            x = rnd(300, K, seed=2).to(dev)
            y = torch.empty(300, N, device=dev)
            ops.linear_fwd(x, w, y, None)
            assert not last_kernel(L).startswith("gemm_panel"), (mode, N, K, last_kernel(L))
            assert err(y, x.double().cpu() @ w.double().cpu().t()) < (tol(K) if mode != "bf16" else 1.0)
        monkeypatch.setenv("NACF_GEMM_MODE", "bf16x3")
        x = rnd(300, 512, seed=2).to(dev)
        w = rnd(256, 512, seed=9).to(dev)           # not registered
        y = torch.empty(300, 256, device=dev)
        ops.linear_fwd(x, w, y, None)
        assert not last_kernel(L).startswith("gemm_panel")
        # whole 32-row blocks of a registered matrix are served (the packed q|k|v weights used slice-wise) ...
        y = torch.empty(300, 128, device=dev)
        ops.linear_fwd(x, W.w[3][128:], y, None)
        assert last_kernel(L).startswith("gemm_panel"), last_kernel(L)
        assert err(y, x.double().cpu() @ W.w[3][128:].double().cpu().t()) < tol(512)
    finally:
        W.close()


def test_panel_fused_epilogue_rows_and_dropout(dev, monkeypatch):
    """every field of the nn.Linear epilogue, a live-row list with dead-row fill (ragged last panel, more panels than one
    workgroup walks), accumulation into dX, and dropout masks equal to the other kernels'"""
    ops, L = _ops()
    monkeypatch.setenv("NACF_GEMM_MODE", "bf16x3")
    monkeypatch.setenv("NACF_GEMM_PANEL", "1")
    M, N, K = 5000, 512, 512
    W = Weights(ops, dev, [(N, K)], scale=0.3)
    try:
        w = W.w[0]
        x, b, r = rnd(M, K, seed=4), rnd(N, seed=6), rnd(M, N, seed=7)
        tok = torch.randint(0, 3, (M,), generator=torch.Generator().manual_seed(5))
        z = x.double() @ w.double().cpu().t() + b.double()
        gelu = 0.5 * z * (1 + torch.tanh(math.sqrt(2 / math.pi) * (z + 0.044715 * z ** 3)))
        ref = (gelu + r.double()) * tok.ne(0).double().unsqueeze(1)
        y, pre = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
        ops.linear_fwd(x.to(dev), w, y, ops.Epi(bias=b.to(dev), act=L.ACT_GELU_NEW, preact=pre, residual=r.to(dev),
                                                row_tokens=tok.to(dev)))
        assert last_kernel(L).startswith("gemm_panel_kernel")
        assert err(pre, z) < tol(K) and err(y, ref) < tol(K)
        # live rows: dead rows are never read and zero-filled on request
        live = tok.ne(PAD).nonzero().squeeze(1)
        xn = x.clone()
        xn[tok.eq(PAD)] = float("nan")
        rows = ops.rowset_build(tokens=tok.to(dev))
        y = torch.full((M, N), 7.0, device=dev)
        ops.linear_fwd(xn.to(dev), w, y, ops.Epi(bias=b.to(dev)), rows, zero_dead=True)
        assert last_kernel(L).startswith("gemm_panel_kernel")
        assert err(y[live.to(dev)], x[live].double() @ w.double().cpu().t() + b.double()) < tol(K)
        assert float(y[tok.eq(PAD).to(dev)].abs().max()) == 0.0
        dz = rnd(M, N, seed=8)
        dz[tok.eq(PAD)] = float("nan")
        dx = torch.full((M, K), 7.0, device=dev)
        ops.linear_bwd_data(dz.to(dev), w, dx, rows=rows, zero_dead=True)
        assert last_kernel(L).startswith("gemm_panel_kernel<2, 4, 3, EpiStore")
        assert err(dx[live.to(dev)], dz[live].double() @ w.double().cpu()) < tol(N)
        assert float(dx[tok.eq(PAD).to(dev)].abs().max()) == 0.0
        # without the fill the dead rows keep what they held
        y = torch.full((M, N), 7.0, device=dev)
        ops.linear_fwd(xn.to(dev), w, y, ops.Epi(bias=b.to(dev)), rows, zero_dead=False)
        assert float((y[tok.eq(PAD).to(dev)] - 7.0).abs().max()) == 0.0
        # accumulate (beta = 1)
        base = rnd(M, K, seed=9).to(dev)
        dz2 = rnd(M, N, seed=10).to(dev)
        acc = base.clone()
        ops.linear_bwd_data(dz2, w, acc, beta=1.0)
        assert last_kernel(L).startswith("gemm_panel_kernel")
        assert err(acc, base.double().cpu() + dz2.double().cpu() @ w.double().cpu()) < tol(N)
        # dropout: same masks as the other kernels
        rng = ops.RngState(1234, dev)
        outs = {}
        for panel in ("1", "0"):
            monkeypatch.setenv("NACF_GEMM_PANEL", panel)
            yd = torch.empty(M, N, device=dev)
            ops.linear_fwd(x.to(dev), w, yd, ops.Epi(bias=b.to(dev), p1=0.5, salt1=11, residual=r.to(dev), p2=0.25, salt2=12, rng=rng))
            outs[panel] = yd
        assert torch.equal(outs["1"].eq(0), outs["0"].eq(0))
        assert err(outs["1"], outs["0"]) < tol(K, 2.0) and bool(outs["1"].ne(0).any())
    finally:
        W.close()


def test_panel_is_deterministic_and_independent_of_the_grid(dev, monkeypatch):
    """the four waves' partial sums are added in wave order: two runs, and the same rows computed as part of a larger
    launch (other tiles per workgroup, other neighbours), are bit-identical"""
    ops, L = _ops()
    monkeypatch.setenv("NACF_GEMM_MODE", "bf16x3")
    monkeypatch.setenv("NACF_GEMM_PANEL", "1")
    W = Weights(ops, dev, [(1536, 512)], scale=0.3)
    try:
        w = W.w[0]
        x = rnd(6000, 512, seed=1).to(dev)
        y1, y2, y3 = (torch.empty(6000, 1536, device=dev) for _ in range(3))
        ops.linear_fwd(x, w, y1, None)
        ops.linear_fwd(x, w, y2, None)
        assert torch.equal(y1, y2)
        ops.linear_fwd(x[:640], w, y3[:640], None)
        assert last_kernel(L).startswith("gemm_panel_kernel")
        assert torch.equal(y1[:640], y3[:640])
    finally:
        W.close()


def test_heuristic_takes_the_panel_kernel_on_the_long_reduce_launches_of_the_decoder(dev, monkeypatch):
    """the shape rule (NACF_GEMM_PANEL=2; the kernel is off by default): FFN2 forward / FFN1 dX / q|k|v dX at B = 128 (5120 slots,
    live-row list) run on the panel kernel"""
    ops, L = _ops()
    monkeypatch.setenv("NACF_GEMM_MODE", "bf16x3")
    monkeypatch.setenv("NACF_GEMM_PANEL", "2")
    W = Weights(ops, dev, [(512, 2048), (2048, 512), (1536, 512)], scale=0.2)
    try:
        tok = (torch.rand(5120, generator=torch.Generator().manual_seed(3)) < 0.5).long().to(dev)
        rows = ops.rowset_build(tokens=tok)
        x = rnd(5120, 2048, seed=1).to(dev)
        y = torch.empty(5120, 512, device=dev)
        ops.linear_fwd(x, W.w[0], y, None, rows, zero_dead=True)
        assert last_kernel(L).startswith("gemm_panel_kernel"), last_kernel(L)
        dx = torch.empty(5120, 512, device=dev)
        ops.linear_bwd_data(x, W.w[1], dx, rows=rows, zero_dead=True)
        assert last_kernel(L).startswith("gemm_panel_kernel"), last_kernel(L)
        dz = rnd(5120, 1536, seed=2).to(dev)
        ops.linear_bwd_data(dz, W.w[2], dx, rows=rows, zero_dead=True)
        assert last_kernel(L).startswith("gemm_panel_kernel"), last_kernel(L)
        # ... and nothing else: a short reduce dimension, or no row list (the encoder's launches belong to the wide kernel)
        y2 = torch.empty(5120, 2048, device=dev)
        ops.linear_fwd(dx, W.w[1], y2, None, rows, zero_dead=True)
        assert not last_kernel(L).startswith("gemm_panel_kernel"), last_kernel(L)
        xe = rnd(7680, 2048, seed=4).to(dev)
        ye = torch.empty(7680, 512, device=dev)
        ops.linear_fwd(xe, W.w[0], ye, None)
        assert not last_kernel(L).startswith("gemm_panel_kernel"), last_kernel(L)
        # off by default
        monkeypatch.delenv("NACF_GEMM_PANEL")
        ops.linear_fwd(x, W.w[0], y, None, rows, zero_dead=True)
        assert not last_kernel(L).startswith("gemm_panel_kernel"), last_kernel(L)
    finally:
        W.close()
