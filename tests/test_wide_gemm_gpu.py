"""GPU: the wide-wave-tile exact-mode GEMM (csrc/gemm_bf16_wide.hpp: one workgroup per CU, 64 x 128 wave tiles, DMA staging,
v_mfma_f32_32x32x16_bf16) behind nacf_linear_fwd / nacf_linear_bwd_data, and the grouped launch of independent problems.

Bars: fp64 reference at the fp32 kernels' tolerance (the kernel is a drop-in for the parity path); against the 128x128 /
64x64 kernels of the same mode only rounding differs (another matrix-instruction shape sums in another order); dropout
masks are functions of the element index, so they are IDENTICAL to the other kernels'; a grouped launch is bit-identical
to the same problems launched one by one.
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
    """a flat buffer of weight matrices with registered three-plane images (the wide kernel reads images only)"""

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


SHAPES = [(1000, 512, 256), (700, 300, 128), (333, 1030, 192), (600, 256, 2048), (129, 257, 320), (64, 256, 128)]


@pytest.mark.parametrize("mt", ["1", "2"])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_wide_forward_and_dx_vs_fp64_and_the_other_kernels(dev, M, N, K, mt, monkeypatch):
    ops, L = _ops()
    monkeypatch.setenv("NACF_GEMM_MODE", "bf16x3")
    W = Weights(ops, dev, [(N, K)])
    try:
        w = W.w[0]
        x, b = rnd(M, K, seed=1).to(dev), rnd(N, seed=3).to(dev)
        dz = rnd(M, N + (-N) % 4, seed=4).to(dev)[:, :N]
        ref_y = x.double().cpu() @ w.double().cpu().t() + b.double().cpu()
        ref_dx = dz.double().cpu() @ w.double().cpu()
        out = {}
        for wide in (mt, "0"):
            monkeypatch.setenv("NACF_GEMM_WIDE", wide)
            y, dx = torch.full((M, N), 7.0, device=dev), torch.full((M, K), 7.0, device=dev)
            ops.linear_fwd(x, w, y, ops.Epi(bias=b))
            k1 = last_kernel(L)
            ops.linear_bwd_data(dz, w, dx)
            k2 = last_kernel(L)
            out[wide] = (y, dx, k1, k2)
        y, dx, k1, k2 = out[mt]
        assert k1.startswith("gemm_wide_kernel<%s, EpiLinear" % mt), k1
        if N % 64 == 0 and N >= 128:          # dX reduces over N: whole k-tiles in an even number >= 4
            assert k2.startswith("gemm_wide_kernel<%s, EpiStore" % mt), k2
        else:
            assert k2.startswith("gemm_bf16_kernel"), k2
        assert out["0"][2].startswith("gemm_bf16_kernel") and out["0"][3].startswith("gemm_bf16_kernel")
        assert err(y, ref_y) < tol(K) and err(dx, ref_dx) < tol(N)
        assert err(y, out["0"][0]) < tol(K, 0.2) and err(dx, out["0"][1]) < tol(N, 0.2)
    finally:
        W.close()


@pytest.mark.parametrize("mt", ["1", "2"])
def test_wide_is_not_taken_when_not_eligible(dev, mt, monkeypatch):
    """odd k-tile counts, fewer than four k-tiles, no registered image, the other modes: the 2-per-CU kernels run"""
    ops, L = _ops()
    monkeypatch.setenv("NACF_GEMM_WIDE", mt)
    W = Weights(ops, dev, [(256, 96), (256, 160), (256, 64), (256, 256)])
    try:
        for mode, w in [("bf16x3", W.w[0]), ("bf16x3", W.w[1]), ("bf16x3", W.w[2]), ("bf16", W.w[3]), ("f32", W.w[3])]:
            monkeypatch.setenv("NACF_GEMM_MODE", mode)
            N, K = w.shape
            x = rnd(300, K, seed=2).to(dev)
            y = torch.empty(300, N, device=dev)
            ops.linear_fwd(x, w, y, None)
            assert not last_kernel(L).startswith("gemm_wide"), (mode, K, last_kernel(L))
            assert err(y, x.double().cpu() @ w.double().cpu().t()) < (tol(K) if mode != "bf16" else 1.0)
        monkeypatch.setenv("NACF_GEMM_MODE", "bf16x3")
        w = rnd(256, 256, seed=9).to(dev)           # not registered: converted inside the (other) kernel
        x = rnd(300, 256, seed=2).to(dev)
        y = torch.empty(300, 256, device=dev)
        ops.linear_fwd(x, w, y, None)
        assert last_kernel(L).startswith("gemm_bf16_kernel")
    finally:
        W.close()


@pytest.mark.parametrize("mt", ["1", "2"])
def test_wide_fused_epilogue_rows_and_dropout(dev, mt, monkeypatch):
    """every field of the nn.Linear epilogue, a live-row list with dead-row fill, and dropout masks equal to the other
    kernels' (they depend on the element index only)"""
    ops, L = _ops()
    monkeypatch.setenv("NACF_GEMM_MODE", "bf16x3")
    M, N, K = 900, 512, 256
    W = Weights(ops, dev, [(N, K)], scale=0.3)
    try:
        w = W.w[0]
        x, b, r = rnd(M, K, seed=4), rnd(N, seed=6), rnd(M, N, seed=7)
        tok = torch.randint(0, 3, (M,), generator=torch.Generator().manual_seed(5))
        z = x.double() @ w.double().cpu().t() + b.double()
        gelu = 0.5 * z * (1 + torch.tanh(math.sqrt(2 / math.pi) * (z + 0.044715 * z ** 3)))
        ref = (gelu + r.double()) * tok.ne(0).double().unsqueeze(1)
        monkeypatch.setenv("NACF_GEMM_WIDE", mt)
        y, pre = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
        ops.linear_fwd(x.to(dev), w, y, ops.Epi(bias=b.to(dev), act=L.ACT_GELU_NEW, preact=pre, residual=r.to(dev),
                                                row_tokens=tok.to(dev)))
        assert last_kernel(L).startswith("gemm_wide_kernel<%s" % mt)
        assert err(pre, z) < tol(K) and err(y, ref) < tol(K)
        # live rows: dead rows are never read and zero-filled on request
        live = tok.ne(PAD).nonzero().squeeze(1)
        xn = x.clone()
        xn[tok.eq(PAD)] = float("nan")
        rows = ops.rowset_build(tokens=tok.to(dev))
        y = torch.full((M, N), 7.0, device=dev)
        ops.linear_fwd(xn.to(dev), w, y, ops.Epi(bias=b.to(dev)), rows, zero_dead=True)
        assert last_kernel(L).startswith("gemm_wide_kernel<%s" % mt)
        assert err(y[live.to(dev)], x[live].double() @ w.double().cpu().t() + b.double()) < tol(K)
        assert float(y[tok.eq(PAD).to(dev)].abs().max()) == 0.0
        dz = rnd(M, N, seed=8)
        dz[tok.eq(PAD)] = float("nan")
        dx = torch.full((M, K), 7.0, device=dev)
        ops.linear_bwd_data(dz.to(dev), w, dx, rows=rows, zero_dead=True)
        assert last_kernel(L).startswith("gemm_wide_kernel<%s, EpiStore" % mt)
        assert err(dx[live.to(dev)], dz[live].double() @ w.double().cpu()) < tol(N)
        assert float(dx[tok.eq(PAD).to(dev)].abs().max()) == 0.0
        # accumulate (beta = 1), as the HighWay backward does
        base = rnd(M, K, seed=9).to(dev)
        dz2 = rnd(M, N, seed=10).to(dev)
        acc = base.clone()
        ops.linear_bwd_data(dz2, w, acc, beta=1.0)
        assert err(acc, base.double().cpu() + dz2.double().cpu() @ w.double().cpu()) < tol(N)
        # dropout: same masks as the other kernels
        rng = ops.RngState(1234, dev)
        outs = {}
        for wide in (mt, "0"):
            monkeypatch.setenv("NACF_GEMM_WIDE", wide)
            yd = torch.empty(M, N, device=dev)
            ops.linear_fwd(x.to(dev), w, yd, ops.Epi(bias=b.to(dev), p1=0.5, salt1=11, residual=r.to(dev), p2=0.25, salt2=12, rng=rng))
            outs[wide] = yd
        kept_a, kept_b = outs[mt].ne(r.to(dev) * 0), outs["0"].ne(r.to(dev) * 0)
        assert torch.equal(outs[mt].eq(0), outs["0"].eq(0))
        assert err(outs[mt], outs["0"]) < tol(K, 2.0) and bool(kept_a.any()) and bool(kept_b.any())
    finally:
        W.close()


def test_wide_splitk_dx(dev, monkeypatch):
    """dX over a long reduce dimension with reduce splits (the slab + combine path) on the wide kernel"""
    ops, L = _ops()
    monkeypatch.setenv("NACF_GEMM_MODE", "bf16x3")
    monkeypatch.setenv("NACF_GEMM_WIDE", "2")
    monkeypatch.setenv("NACF_GEMM_SPLITS", "4")
    M, N, K = 300, 4096, 256
    W = Weights(ops, dev, [(N, K)], scale=0.1)
    try:
        dz = rnd(M, N, seed=3).to(dev)
        dx = torch.empty(M, K, device=dev)
        ops.linear_bwd_data(dz, W.w[0], dx)
        assert last_kernel(L).startswith("gemm_wide_kernel<2, EpiStore"), last_kernel(L)
        assert err(dx, dz.double().cpu() @ W.w[0].double().cpu()) < tol(N, 0.2)
    finally:
        W.close()


def test_wide_group_is_bit_identical_to_single_launches(dev, monkeypatch):
    ops, L = _ops()
    monkeypatch.setenv("NACF_GEMM_MODE", "bf16x3")
    monkeypatch.setenv("NACF_GEMM_WIDE", "2")
    mats = [(512, 2048), (512, 2048), (256, 512), (1024, 512)]
    W = Weights(ops, dev, mats, scale=0.2)
    try:
        xs = [rnd(900 + 64 * i, K, seed=20 + i).to(dev) for i, (N, K) in enumerate(mats)]
        bs = [rnd(N, seed=30 + i).to(dev) for i, (N, K) in enumerate(mats)]
        dzs = [rnd(x.shape[0], N, seed=40 + i).to(dev) for i, (x, (N, K)) in enumerate(zip(xs, mats))]

        def run(grouped):
            ys = [torch.full((x.shape[0], N), 3.0, device=dev) for x, (N, K) in zip(xs, mats)]
            dxs = [torch.full((x.shape[0], K), 3.0, device=dev) for x, (N, K) in zip(xs, mats)]
            if grouped:
                with ops.wide_group():
                    for x, w, b, y in zip(xs, W.w, bs, ys):
                        ops.linear_fwd(x, w, y, ops.Epi(bias=b))
                    for dz, w, dx in zip(dzs, W.w, dxs):
                        ops.linear_bwd_data(dz, w, dx)
                    torch.cuda.synchronize()
                    assert all(float(y.flatten()[0]) == 3.0 for y in ys)         # nothing launched yet
                assert last_kernel(L).startswith("gemm_wide_group_kernel<2, EpiStore"), last_kernel(L)
            else:
                for x, w, b, y in zip(xs, W.w, bs, ys):
                    ops.linear_fwd(x, w, y, ops.Epi(bias=b))
                    assert last_kernel(L).startswith("gemm_wide_kernel<2, EpiLinear")
                for dz, w, dx in zip(dzs, W.w, dxs):
                    ops.linear_bwd_data(dz, w, dx)
            return ys, dxs

        y1, dx1 = run(False)
        y2, dx2 = run(True)
        for a, b in zip(y1 + dx1, y2 + dx2):
            assert torch.equal(a, b)
        for x, w, b, y in zip(xs, W.w, bs, y2):
            assert err(y, x.double().cpu() @ w.double().cpu().t() + b.double().cpu()) < tol(x.shape[1], 0.5)
    finally:
        W.close()


def test_joint_encoder_streams_equal_the_per_stream_nodes(dev, monkeypatch):
    """models/Encoder.py: layer-by-layer over the modalities (grouped GEMMs) == one autograd node per modality, bit for bit"""
    import nacf_amd
    from nacf_amd import synthetic as S
    monkeypatch.setenv("NACF_GEMM_MODE", "bf16x3")
    monkeypatch.setenv("NACF_GEMM_WIDE", "2")       # every eligible GEMM on the wide kernel: grouped (joint) vs one by one
    res = {}
    for joint in (True, False):
        opt = nacf_amd.opts.make_opt("NACF", "MSRVTT", with_category=True, max_len=20, vocab_size=300, fused_loss=True,
                                     hidden_dropout_prob=0.5, encoder_dropout=0.5, encoder_joint_streams=joint)
        sd = S.init_state_dict(opt, seed=0)
        model = nacf_amd.get_model(opt)
        model.load_state_dict({k: v.clone() for k, v in sd.items()})
        model.to(dev).train()
        b = S.synth_batch(opt, 24, 60, seed=1)
        feats = [f.to(dev) for f in b["feats"]]
        outs, _ = model.encoder(feats)
        loss = sum((o * (i + 1)).sum() for i, o in enumerate(outs))
        loss.backward()
        res[joint] = ([o.detach().clone() for o in outs], {n: p.grad.detach().clone() for n, p in model.encoder.named_parameters()})
    for a, b in zip(res[True][0], res[False][0]):
        assert torch.equal(a, b)
    for n in res[True][1]:
        assert torch.equal(res[True][1][n], res[False][1][n]), n


def test_fixed_register_kernels_selftest_runs_and_leaves_no_trace(dev, monkeypatch):
    """ops.selftest_fixed_register_kernels (once per process before the first exact-mode weight images): wide kernel, both
    wave-tile heights, against the 128x128 kernel; the GEMM mode and the tuning environment are as before afterwards"""
    import os
    from nacf_amd.runtime import ops
    monkeypatch.setenv("NACF_GEMM_MODE", "bf16")          # an override in force while the self-test runs
    before_env = {k: os.environ.get(k) for k in ("NACF_GEMM_TILE", "NACF_GEMM_WIDE", "NACF_GEMM_MODE")}
    ops._FIXED_REG_SELFTEST["done"] = False
    ops.selftest_fixed_register_kernels(dev)
    assert ops._FIXED_REG_SELFTEST["done"]
    assert {k: os.environ.get(k) for k in before_env} == before_env
    assert ops.gemm_mode() == 1                            # the override is back in force
    monkeypatch.delenv("NACF_GEMM_MODE")
    assert ops.gemm_mode() in (0, 1, 3)
