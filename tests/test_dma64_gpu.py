"""GPU: the DMA-fed 32 x 64 kernel of the throughput mode's small launches (csrc/gemm_dma64.hpp) through the C ABI: the same
bits as gemm_bf16_kernel<64, 64> (NACF_DMA64=0) for nn.Linear forward with every epilogue feature, dX (with split-K slabs), the
vocabulary projection with soft-max statistics; live-row lists with dead-row fill, ragged extents."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    import nacf_amd  # noqa: F401
    from nacf_amd.runtime import ops, lib
    return ops, lib


def _images(ops, ws, transposed):
    """one flat buffer + images of the weight matrices `ws` ([N, K] each)"""
    offs, off = [], 0
    for w in ws:
        offs.append(off)
        off += (w.numel() + 3) // 4 * 4
    flat = torch.empty(off, device=ws[0].device)
    views = []
    for o, w in zip(offs, ws):
        flat[o:o + w.numel()] = w.reshape(-1)
        views.append(flat[o:o + w.numel()].view(w.shape))
    imgs = ops.WeightImages(flat, [(o, w.shape[0], w.shape[1], transposed) for o, w in zip(offs, ws)], 1)
    imgs.refresh()
    return views, imgs


def test_dma64_equals_the_64x64_kernel_bit_for_bit(dev, monkeypatch):
    ops, L = _ops()
    monkeypatch.setenv("NACF_GEMM_MODE", "bf16")
    monkeypatch.setenv("NACF_GEMM_TILE", "64")
    lib = L.load()
    g = torch.Generator().manual_seed(7)
    r = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).to(dev)
    shapes = [(1280, 512, 512), (300, 101, 64), (2970, 512, 2048), (77, 64, 96), (640, 1536, 512)]      # [M, N, K]
    ws, imgs = _images(ops, [r(N, K) for _, N, K in shapes], True)
    tok = (torch.rand(1280, generator=g) < 0.55).long().to(dev)
    rows = ops.rowset_build(tokens=tok)
    rng = ops.RngState(11, dev)
    vocab_w, vimg = _images(ops, [r(1000, 128)], False)

    def run(flag):
        monkeypatch.setenv("NACF_DMA64", flag)
        outs, names = [], []
        for (M, N, K), w in zip(shapes, ws):
            x = r(M, K)
            y, pre = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
            epi = ops.Epi(bias=r(N), act=L.ACT_RELU, preact=pre, residual=r(M, N) if N % 4 == 0 else None)
            ops.linear_fwd(x, w, y, epi)
            names.append(lib.nacf_gemm_last_kernel().decode())
            dz = r(M, (N + 3) // 4 * 4)[:, :N]
            dx = torch.empty(M, K, device=dev)
            ops.linear_bwd_data(dz, w, dx)
            names.append(lib.nacf_gemm_last_kernel().decode())
            outs += [y, pre, dx]
        # live rows + dead-row fill, dropout and the <pad> row mask in the epilogue
        M, N, K = shapes[0]
        x, y = r(M, K), torch.full((M, N), 7.0, device=dev)
        epi = ops.Epi(bias=r(N), act=L.ACT_GELU_NEW, p1=0.3, salt1=5, residual=r(M, N), p2=0.2, salt2=9, row_tokens=tok, rng=rng)
        ops.linear_fwd(x, ws[0], y, epi, rows=rows, zero_dead=True)
        names.append(lib.nacf_gemm_last_kernel().decode())
        dx = torch.full((M, K), 7.0, device=dev)
        ops.linear_bwd_data(r(M, N), ws[0], dx, rows=rows, zero_dead=True)
        outs += [y, dx]
        # the vocabulary projection with soft-max statistics out of the epilogue
        h, lab = r(600, 128), torch.randint(1, 1000, (600,), generator=g).to(dev)
        buf = torch.empty(600, ops.vocab_ld(1000), device=dev)
        lse, llp, am = torch.empty(600, device=dev), torch.empty(600, device=dev), torch.empty(600, dtype=torch.int64, device=dev)
        ops.vocab_lse_fwd(h, vocab_w[0], r(1000), buf[:, :1000], lab, lse, am, llp, None)
        names.append(lib.nacf_gemm_last_kernel().decode())
        outs += [buf[:, :1000].clone(), lse, llp, am]
        torch.cuda.synchronize()
        return outs, names

    g.manual_seed(7)
    new, names_new = run("1")
    g.manual_seed(7)
    old, names_old = run("0")
    # (not every launch is eligible: a reduce dimension that is no multiple of 32, more than 768 tiles)
    assert sum("gemm_dma64_kernel" in n for n in names_new) >= 8 and "gemm_dma64_kernel<32, EpiArgmax>" in names_new[-1], names_new
    assert "gemm_dma64_kernel" in names_new[-2], names_new      # the row-list launch
    assert not any("gemm_dma64_kernel" in n for n in names_old), names_old
    for i, (a, b) in enumerate(zip(new, old)):
        assert torch.equal(a, b), i
    assert float(new[-6].abs().max()) > 0 and bool((new[-6][tok == 0] == 0).all())      # dead rows of the row-list launch: zero-filled
