"""GPU: the kernels on the 256 x 256 x 64 eight-phase bf16-resident GEMM body (csrc/gemm_g256.hpp, nacf_gemm_g256.hip),
through the C ABI.  References: fp64 on the bf16-ROUNDED operands (what the throughput mode multiplies; the rounding itself
is the mode's stated tolerance, tests/test_bf16_mode_gpu.py)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    import nacf_amd  # noqa: F401
    from nacf_amd.runtime import ops, lib
    return ops, lib


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float64)


SHAPES = [(5120, 512, 512), (5120, 2048, 512), (5120, 512, 2048), (7680, 512, 2048), (5120, 1536, 512), (15360, 1024, 512),
          (5120, 10547, 512), (700, 130, 260), (64, 256, 256), (100, 384, 136), (3000, 520, 1000)]


@pytest.mark.parametrize("mode", ["bf16", "bf16x3"])
def test_dw_group_g256_against_fp64(dev, monkeypatch, mode):
    """Weight gradients on the 256 x 256 eight-phase body for fp32 operands (csrc/gemm_g256w.hpp): ONE grouped launch (live-row
    gather in the DMA, register transpose + bf16 split of the fragments, round-robin k-splits, bias gradient from the raw
    fragments) + the combine of the split problems.  Row lists (~58 % live, and an EMPTY one), bias gradients, beta = 1, ragged
    extents (N = 10547, 130 x 260), fewer rows than a k-tile; run-to-run bit-identical; as accurate as round 4's grouped
    128 x 128 kernel (NACF_DW_G256=0).  Exact mode: fp64 of the fp32 operands; throughput mode: fp64 of the bf16-rounded ones."""
    ops, L = _ops()
    monkeypatch.setenv("NACF_GEMM_MODE", mode)
    g = torch.Generator().manual_seed(5)
    r = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).to(dev)
    probs = []
    for i, (M, N, K) in enumerate(SHAPES):
        dz, x = r(M, ops.vocab_ld(N))[:, :N], r(M, K)
        rows = None
        if i in (0, 1, 6, 7):
            tok = (torch.rand(M, generator=g) < 0.58).to(dev).long()
            rows = ops.rowset_build(tokens=tok)
        if i == 9:
            rows = ops.rowset_build(tokens=torch.zeros(M, dtype=torch.long, device=dev))      # nothing is live
        probs.append((dz, x, r(N, K), r(N) if i != 2 else None, rows))
    lib = L.load()

    def run(flag):
        monkeypatch.setenv("NACF_DW_G256", flag)
        for _ in range(2):                                    # (the first pass sizes the group's buffer)
            out = [(w0.clone(), None if b0 is None else b0.clone()) for _, _, w0, b0, _ in probs]
            with ops.dw_group():
                for (dz, x, _, _, rows), (dw, db) in zip(probs, out):
                    ops.linear_bwd_weight(dz, x, dw, db, beta=1.0, rows=rows)
            torch.cuda.synchronize()
        name = (lib.nacf_gemm_last_kernel() or b"").decode()
        return out, name

    new, name = run("3")
    assert "g256" in name, name
    new2, _ = run("3")
    old, name_old = run("0")
    assert "g256" not in name_old, name_old
    cast = _bf if mode == "bf16" else (lambda t: t.double())
    for (dz, x, w0, b0, rows), (a, ab), (a2, ab2), (o, ob) in zip(probs, new, new2, old):
        assert torch.equal(a, a2) and (ab is None or torch.equal(ab, ab2))
        dz64, x64, dzf = cast(dz), cast(x), dz.double()
        if rows is not None:
            live = rows.rows[:int(rows.count)].long()
            dz64, x64, dzf = dz64[live], x64[live], dzf[live]
        ref = w0.double() + dz64.t() @ x64
        scale = max(float(ref.abs().max()), 1.0)
        e = float((a.double() - ref).abs().max()) / scale
        e_old = float((o.double() - ref).abs().max()) / scale
        assert e <= max(3e-6, 1.5 * e_old), (tuple(dz.shape), e, e_old)      # fp32 accumulation of exact products
        if b0 is not None:
            refb = b0.double() + dzf.sum(0)                    # the bias gradient sums the fp32 values
            assert float((ab.double() - refb).abs().max()) <= 1e-5 * float(refb.abs().max()) + 1e-4
