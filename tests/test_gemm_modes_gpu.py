"""GPU: the bf16 matrix-core GEMM family (csrc/gemm_bf16.hpp) behind the SAME entry points as the fp32 MFMA kernels.

  NACF_GEMM_MODE=bf16x3  exact mode: every fp32 operand split into three bf16 terms, six MFMAs per product block.
                         Bar = the fp32 kernels' bar (fp64 reference, 1e-5 * sqrt(K)-class tolerances): the mode is a
                         drop-in for the parity path, so it must be fp32-accurate, not "bf16-accurate".
  NACF_GEMM_MODE=bf16    throughput mode: operands rounded to bf16.  Bar = 2^-8 relative per operand.

Both operand paths are covered: weights converted inside the kernel (ad-hoc tensors) and weights copied from
pre-split images (ops.WeightImages) -- which must give bit-identical results, the split arithmetic being the same.
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


def tol(mode, K, scale=1.0):
    # exact: fp32 round-off of a K-term sum; bf16: two operands rounded to 8 significant bits
    return (1e-5 * math.sqrt(K) + 1e-5) * scale if mode == "bf16x3" else (6e-3 * math.sqrt(K) + 1e-3) * scale


MODES = ["bf16x3", "bf16"]
SHAPES = [(64, 64, 32), (128, 128, 64), (200, 136, 72), (37, 101, 64), (1, 10, 32), (300, 40, 100), (513, 257, 129),
          (96, 30, 62), (256, 1024, 512), (700, 200, 40)]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("tile", ["64", "128"])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_linear_fwd(dev, M, N, K, tile, mode, monkeypatch):
    ops, _ = _ops()
    monkeypatch.setenv("NACF_GEMM_MODE", mode)
    monkeypatch.setenv("NACF_GEMM_TILE", tile)
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    y = torch.empty(M, N, device=dev)
    ops.linear_fwd(x.to(dev), w.to(dev), y, ops.Epi(bias=b.to(dev)))
    ref = x.double() @ w.double().t() + b.double()
    assert err(y, ref) < tol(mode, K)
    if K % 4 == 0:      # 16-byte addressable operands: the launch really was a bf16 matrix-core kernel
        assert ops.L.load().nacf_gemm_last_kernel().decode().startswith("gemm_bf16_kernel<%s, %s" % (tile, tile))


@pytest.mark.parametrize("mode", MODES)
def test_exact_mode_is_at_least_as_accurate_as_the_fp32_mfma(dev, mode, monkeypatch):
    """K = 512 / 2048 reductions of the model on O(1) data: error vs fp64 of the three kernels side by side"""
    ops, _ = _ops()
    out = {}
    for K in (512, 2048):
        x, w = rnd(512, K, seed=1), rnd(384, K, seed=2)
        ref = x.double() @ w.double().t()
        for m in ("f32", mode):
            monkeypatch.setenv("NACF_GEMM_MODE", m)
            y = torch.empty(512, 384, device=dev)
            ops.linear_fwd(x.to(dev), w.to(dev), y, None)
            out[(K, m)] = err(y, ref)
    print("max |err| vs fp64:", out)
    if mode == "bf16x3":
        for K in (512, 2048):
            assert out[(K, mode)] <= 1.5 * out[(K, "f32")] + 1e-7, out


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("tile", ["64", "128"])
@pytest.mark.parametrize("M,N,K", [(128, 64, 64), (200, 136, 72), (37, 101, 64), (513, 96, 200), (256, 2048, 512),
                                   (1000, 300, 128)])
def test_linear_bwd_data_and_weight(dev, M, N, K, tile, mode, monkeypatch):
    ops, _ = _ops()
    monkeypatch.setenv("NACF_GEMM_MODE", mode)
    monkeypatch.setenv("NACF_GEMM_TILE", tile)
    dz, w, x = rnd(M, N, seed=1), rnd(N, K, seed=2), rnd(M, K, seed=3)
    dx = torch.empty(M, K, device=dev)
    ops.linear_bwd_data(dz.to(dev), w.to(dev), dx)
    assert err(dx, dz.double() @ w.double()) < tol(mode, N)
    base = rnd(M, K, seed=9)
    dx2 = base.to(dev).clone()
    ops.linear_bwd_data(dz.to(dev), w.to(dev), dx2, beta=1.0)
    assert err(dx2, base.double() + dz.double() @ w.double()) < tol(mode, N)
    dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    ops.linear_bwd_weight(dz.to(dev), x.to(dev), dw, db, beta=0.0)
    assert err(dw, dz.double().t() @ x.double()) < tol(mode, M)
    assert err(db, dz.double().sum(0)) < 1e-4 * math.sqrt(M)          # the bias gradient is summed in fp32 in every mode
    dw1, db1 = dw.clone(), db.clone()
    ops.linear_bwd_weight(dz.to(dev), x.to(dev), dw1, db1, beta=1.0)
    assert err(dw1, 2 * dz.double().t() @ x.double()) < 2 * tol(mode, M)
    assert err(db1, 2 * dz.double().sum(0)) < 2e-4 * math.sqrt(M)


@pytest.mark.parametrize("mode", MODES)
def test_splitk_paths(dev, mode, monkeypatch):
    """dX with a long reduce dimension (the vocabulary projection's shape class) and dW over many rows: split-K slabs"""
    ops, _ = _ops()
    monkeypatch.setenv("NACF_GEMM_MODE", mode)
    M, N, K = 384, 4200, 64
    dz, w, x = rnd(M, N, seed=1, scale=0.2), rnd(N, K, seed=2), rnd(M, K, seed=3)
    dx = torch.empty(M, K, device=dev)
    ops.linear_bwd_data(dz.to(dev), w.to(dev), dx)
    assert err(dx, dz.double() @ w.double()) < tol(mode, N, 0.2)
    M, N, K = 6000, 96, 160
    dz, x = rnd(M, N, seed=4), rnd(M, K, seed=5)
    dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    ops.linear_bwd_weight(dz.to(dev), x.to(dev), dw, db, beta=0.0)
    assert err(dw, dz.double().t() @ x.double()) < tol(mode, M)
    assert err(db, dz.double().sum(0)) < 1e-4 * math.sqrt(M)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("M,N,K,tile", [(300, 136, 72, "64"), (1000, 512, 256, "128"), (700, 101, 64, "64"),
                                        (130, 101, 64, "128")])
def test_live_row_gemms(dev, M, N, K, tile, mode, monkeypatch):
    """row sets: only live rows are computed / reduced; dead rows are zero-filled on request and never read"""
    ops, _ = _ops()
    monkeypatch.setenv("NACF_GEMM_MODE", mode)
    monkeypatch.setenv("NACF_GEMM_TILE", tile)
    g = torch.Generator().manual_seed(0)
    tok = torch.randint(0, 3, (M,), generator=g)
    live_idx = tok.ne(PAD).nonzero().squeeze(1)
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    x[tok.eq(PAD)] = float("nan")                          # dead rows must not be read
    rows = ops.rowset_build(tokens=tok.to(dev))
    y = torch.full((M, N), 7.0, device=dev)
    ops.linear_fwd(x.to(dev), w.to(dev), y, ops.Epi(bias=b.to(dev)), rows, zero_dead=True)
    ref = x[live_idx].double() @ w.double().t() + b.double()
    assert err(y[live_idx.to(dev)], ref) < tol(mode, K)
    assert float(y[tok.eq(PAD).to(dev)].abs().max()) == 0.0
    dz = rnd(M, N, seed=4)
    dz[tok.eq(PAD)] = float("nan")
    dx = torch.full((M, K), 7.0, device=dev)
    ops.linear_bwd_data(dz.to(dev), w.to(dev), dx, rows=rows, zero_dead=True)
    assert err(dx[live_idx.to(dev)], dz[live_idx].double() @ w.double()) < tol(mode, N)
    assert float(dx[tok.eq(PAD).to(dev)].abs().max()) == 0.0
    dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    ops.linear_bwd_weight(dz.to(dev), x.to(dev), dw, db, beta=0.0, rows=rows)
    assert err(dw, dz[live_idx].double().t() @ x[live_idx].double()) < tol(mode, M)
    assert err(db, dz[live_idx].double().sum(0)) < 1e-4 * math.sqrt(M)


@pytest.mark.parametrize("mode", MODES)
def test_fused_epilogue_and_activations(dev, mode, monkeypatch):
    ops, L = _ops()
    monkeypatch.setenv("NACF_GEMM_MODE", mode)
    M, N, K = 150, 128, 96
    x, w, b, r = rnd(M, K, seed=4), rnd(N, K, seed=5, scale=0.3), rnd(N, seed=6), rnd(M, N, seed=7)
    tok = torch.randint(0, 3, (M,), generator=torch.Generator().manual_seed(5))
    z = x.double() @ w.double().t() + b.double()
    gelu = 0.5 * z * (1 + torch.tanh(math.sqrt(2 / math.pi) * (z + 0.044715 * z ** 3)))
    ref = (gelu + r.double()) * tok.ne(0).double().unsqueeze(1)
    y, pre = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    ops.linear_fwd(x.to(dev), w.to(dev), y, ops.Epi(bias=b.to(dev), act=L.ACT_GELU_NEW, preact=pre, residual=r.to(dev),
                                                    row_tokens=tok.to(dev)))
    assert err(pre, z) < tol(mode, K) and err(y, ref) < tol(mode, K)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("rows,V,K", [(57, 101, 64), (300, 1000, 64), (600, 10547, 512)])
def test_vocab_argmax_and_lse(dev, rows, V, K, mode, monkeypatch):
    """fused projection + soft-max statistics: decode form (argmax / max prob) and training form (logits, lse, log p)"""
    ops, _ = _ops()
    monkeypatch.setenv("NACF_GEMM_MODE", mode)
    h, w = rnd(rows, K, seed=1), rnd(V, K, seed=2, scale=0.3)
    logits = h.double() @ w.double().t()
    p = torch.softmax(logits, 1)
    tokens = torch.zeros(rows, dtype=torch.int64, device=dev)
    probs = torch.zeros(rows, device=dev)
    ops.vocab_argmax(h.to(dev), w.to(dev), None, None, 0, None, tokens, probs)
    top2 = logits.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > (1e-4 if mode == "bf16x3" else 0.2)
    assert safe.float().mean() > (0.9 if mode == "bf16x3" else 0.2)
    assert torch.equal(tokens.cpu()[safe], logits.argmax(1)[safe])
    if mode == "bf16x3":
        assert err(probs, p.max(1).values) < 2e-5
    lab = torch.randint(1, V, (rows,), generator=torch.Generator().manual_seed(3))
    buf = torch.empty(rows, ops.vocab_ld(V), device=dev)
    lse, llp = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    am = torch.empty(rows, dtype=torch.int64, device=dev)
    ops.vocab_lse_fwd(h.to(dev), w.to(dev), None, buf[:, :V], lab.to(dev), lse, am, llp, None)
    assert err(buf[:, :V], logits) < tol(mode, K, 0.3)
    assert err(lse, torch.logsumexp(logits, 1)) < (2e-5 if mode == "bf16x3" else 2e-2)
    assert err(llp, torch.log_softmax(logits, 1).gather(1, lab.unsqueeze(1)).squeeze(1)) < tol(mode, K, 0.3) + 1e-4


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("tile", ["64", "128"])
def test_weight_images_equal_in_kernel_conversion_bit_for_bit(dev, mode, tile, monkeypatch):
    """pre-split images (row-major for the forward GEMM, transposed for dX) hold exactly the bf16 terms the kernel
    would compute itself, so both operand paths must give IDENTICAL results -- incl. packed / sliced weights, a
    reduce dimension that is not a multiple of 8 (transposed image padded with zeros) and refresh after an update"""
    ops, L = _ops()
    monkeypatch.setenv("NACF_GEMM_MODE", mode)
    monkeypatch.setenv("NACF_GEMM_TILE", tile)
    ns = 3 if mode == "bf16x3" else 1
    mats = [(384, 128), (101, 64), (10, 64), (200, 2048)]          # [N, K]; 101 and 10: ragged transposed images
    offs, off = [], 0
    for N, K in mats:
        offs.append(off)
        off += (N * K + 3) // 4 * 4
    flat = rnd(off, seed=11, scale=0.5).to(dev)
    ws = [flat[o:o + N * K].view(N, K) for o, (N, K) in zip(offs, mats)]

    def run():
        outs = []
        for w in ws:
            N, K = w.shape
            x = rnd(300, K, seed=N).to(dev)
            dzb = rnd(300, (N + 3) // 4 * 4, seed=K + N).to(dev)      # 16-byte row pitch, as the model's padded logits
            dz = dzb[:, :N]
            y, dx = torch.empty(300, N, device=dev), torch.empty(300, K, device=dev)
            ops.linear_fwd(x, w, y, None)
            k1 = L.load().nacf_gemm_last_kernel().decode()
            ops.linear_bwd_data(dz, w, dx)
            k2 = L.load().nacf_gemm_last_kernel().decode()
            outs.append((y, dx, k1, k2))
        half = ws[0][128:256]                                       # a row slice of a registered matrix (q|k|v style)
        x = rnd(77, 128, seed=5).to(dev)
        yh = torch.empty(77, 128, device=dev)
        ops.linear_fwd(x, half, yh, None)
        return outs, yh

    plain, plain_half = run()
    assert all(", 0, 0, %d," % ns in k1 and ", 0, 1, %d," % ns in k2 for _, _, k1, k2 in plain)     # fp32 sources
    imgs = ops.WeightImages(flat, [(o, N, K, True) for o, (N, K) in zip(offs, mats)], ns)
    try:
        imgs.refresh()
        with_img, img_half = run()
        # image sources (the throughput mode's small launches take the DMA-fed kernel, which only exists on images)
        from_img = lambda k: ", 0, 2, %d," % ns in k or (ns == 1 and "gemm_dma64_kernel" in k)
        assert all(from_img(k1) and from_img(k2) for _, _, k1, k2 in with_img)
        for (y0, dx0, _, _), (y1, dx1, _, _) in zip(plain, with_img):
            assert torch.equal(y0, y1) and torch.equal(dx0, dx1)
        assert torch.equal(plain_half, img_half)
        # weights change -> stale until refreshed
        flat.mul_(1.25)
        stale, _ = run()
        imgs.refresh()
        fresh, _ = run()
        assert not torch.equal(stale[0][0], fresh[0][0])
        imgs.close()
        again, _ = run()
        assert torch.equal(again[0][0], fresh[0][0]) and torch.equal(again[3][1], fresh[3][1])
    finally:
        imgs.close()
