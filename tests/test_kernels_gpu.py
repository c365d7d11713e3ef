"""GPU: every HIP entry point vs a plain PyTorch reference of the same op
(fp64 on the host), through the C ABI (nacf_amd.runtime.ops -> ctypes).
Tolerances: fp32 GEMMs 2e-5 * sqrt(K)-ish absolute on O(1) data (stated per
test); index / integer outputs bit-exact."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

PAD, MASK, VIS = 0, 4, 5


def _ops():
    import nacf_amd  # noqa: F401
    from nacf_amd.runtime import ops, lib
    return ops, lib


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def err(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (128, 128, 64), (200, 136, 72), (37, 101, 64), (1, 10, 32),
                                   (300, 40, 100), (513, 257, 129), (96, 30, 62), (256, 1024, 512)])
@pytest.mark.parametrize("tile", ["64", "128"])
def test_linear_fwd_plain(dev, M, N, K, tile, monkeypatch):
    ops, _ = _ops()
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    monkeypatch.setenv("NACF_GEMM_TILE", tile)   # read per call by the library; restored after the test
    y = torch.empty(M, N, device=dev)
    ops.linear_fwd(x.to(dev), w.to(dev), y, ops.Epi(bias=b.to(dev)))
    ref = x.double() @ w.double().t() + b.double()
    assert err(y, ref) < 1e-5 * math.sqrt(K) + 1e-5


def test_linear_fwd_strided_views(dev):
    ops, _ = _ops()
    # operands / outputs that are column slices of wider buffers (packed q|k|v, padded logits)
    M, N, K = 130, 72, 64
    xb, wb = rnd(M, K + 8, seed=1).to(dev), rnd(N, K + 4, seed=2).to(dev)
    yb = torch.zeros(M, N + 12, device=dev)
    ops.linear_fwd(xb[:, 4:4 + K], wb[:, :K], yb[:, 8:8 + N], None)
    ref = xb[:, 4:4 + K].double().cpu() @ wb[:, :K].double().cpu().t()
    assert err(yb[:, 8:8 + N], ref) < 1e-4
    assert float(yb[:, :8].abs().max()) == 0 and float(yb[:, 8 + N:].abs().max()) == 0


@pytest.mark.parametrize("act", ["relu", "gelu_new", "tanh", "sigmoid", "split", "gelu"])
def test_linear_fwd_activations(dev, act):
    ops, L = _ops()
    M, N, K = 150, 128, 96
    x, w, b = rnd(M, K, seed=4), rnd(N, K, seed=5, scale=0.3), rnd(N, seed=6)
    z = x.double() @ w.double().t() + b.double()
    code = {"relu": L.ACT_RELU, "gelu_new": L.ACT_GELU_NEW, "tanh": L.ACT_TANH, "sigmoid": L.ACT_SIGMOID,
            "split": L.ACT_TANH_SIGMOID, "gelu": L.ACT_GELU_ERF}[act]
    ref = {"relu": lambda: F.relu(z),
           "gelu_new": lambda: 0.5 * z * (1 + torch.tanh(math.sqrt(2 / math.pi) * (z + 0.044715 * z ** 3))),
           "tanh": lambda: torch.tanh(z), "sigmoid": lambda: torch.sigmoid(z),
           "split": lambda: torch.cat([torch.tanh(z[:, :64]), torch.sigmoid(z[:, 64:])], 1),
           "gelu": lambda: F.gelu(z)}[act]()
    y = torch.empty(M, N, device=dev)
    pre = torch.empty(M, N, device=dev)
    ops.linear_fwd(x.to(dev), w.to(dev), y, ops.Epi(bias=b.to(dev), act=code, act_split=64, preact=pre))
    assert err(pre, z) < 1e-4
    assert err(y, ref) < 1e-4


def test_linear_fwd_residual_rowmask(dev):
    ops, _ = _ops()
    M, N, K = 96, 64, 64
    x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3), rnd(M, N, seed=4)
    tok = torch.randint(0, 3, (M,), generator=torch.Generator().manual_seed(5))
    y = torch.empty(M, N, device=dev)
    ops.linear_fwd(x.to(dev), w.to(dev), y, ops.Epi(bias=b.to(dev), residual=r.to(dev), row_tokens=tok.to(dev)))
    ref = (x.double() @ w.double().t() + b.double() + r.double()) * tok.ne(0).double().unsqueeze(1)
    assert err(y, ref) < 1e-4
    assert float(y[tok.eq(0).to(dev)].abs().max()) == 0.0   # PAD rows are exactly zero


def test_dropout_statistics_and_backward_mask_identity(dev):
    ops, L = _ops()
    M, N, K = 512, 256, 64
    x, w = rnd(M, K, seed=1).to(dev), rnd(N, K, seed=2).to(dev)
    rng = ops.RngState(1234, dev)
    p = 0.5
    epi = ops.Epi(p1=p, salt1=77, rng=rng)
    y = torch.empty(M, N, device=dev)
    ops.linear_fwd(x, w, y, epi)
    z = torch.empty(M, N, device=dev)
    ops.linear_fwd(x, w, z, None)
    keep = y.ne(0)
    rate = float(keep.float().mean())
    assert abs(rate - (1 - p)) < 0.01, rate                           # keep-rate
    assert err(y[keep], (z / (1 - p))[keep]) < 1e-5                   # 1/(1-p) scaling
    # backward regenerates the identical mask from (seed, step, salt, index)
    dy = torch.ones(M, N, device=dev)
    dz = torch.empty(M, N, device=dev)
    ops.epilogue_bwd(dy, dz, None, epi)
    assert torch.equal(dz.ne(0), keep)
    assert err(dz[keep], torch.full_like(dz[keep], 1 / (1 - p))) < 1e-6
    # a new step gives a new mask; same step gives the same mask
    y2 = torch.empty(M, N, device=dev)
    ops.linear_fwd(x, w, y2, epi)
    assert torch.equal(y2, y)
    rng.advance()
    ops.linear_fwd(x, w, y2, epi)
    assert not torch.equal(y2.ne(0), keep)


@pytest.mark.parametrize("M,N,K", [(128, 64, 64), (200, 136, 72), (37, 101, 64), (513, 96, 200), (256, 2048, 512)])
def test_linear_bwd_data_and_weight(dev, M, N, K):
    ops, _ = _ops()
    dz, w, x = rnd(M, N, seed=1), rnd(N, K, seed=2), rnd(M, K, seed=3)
    dx = rnd(M, K, seed=4).to(dev)
    dx0 = dx.clone()
    ops.linear_bwd_data(dz.to(dev), w.to(dev), dx, beta=1.0)
    ref = dx0.double().cpu() + dz.double() @ w.double()
    assert err(dx, ref) < 1e-5 * math.sqrt(N) + 1e-5
    dw = rnd(N, K, seed=5).to(dev)
    db = rnd(N, seed=6).to(dev)
    dw0, db0 = dw.clone(), db.clone()
    ops.linear_bwd_weight(dz.to(dev), x.to(dev), dw, db, beta=1.0)
    assert err(dw, dw0.double().cpu() + dz.double().t() @ x.double()) < 1e-5 * math.sqrt(M) + 1e-5
    assert err(db, db0.double().cpu() + dz.double().sum(0)) < 1e-5 * math.sqrt(M) + 1e-5
    ops.linear_bwd_weight(dz.to(dev), x.to(dev), dw, None, beta=0.0)
    assert err(dw, dz.double().t() @ x.double()) < 1e-5 * math.sqrt(M) + 1e-5


def test_linear_bwd_weight_splitk_large_m(dev):
    ops, _ = _ops()
    M, N, K = 5120, 64, 128      # many reduce rows, few output tiles -> split-K slabs
    dz, x = rnd(M, N, seed=1), rnd(M, K, seed=2)
    dw = torch.zeros(N, K, device=dev)
    db = torch.zeros(N, device=dev)
    ops.linear_bwd_weight(dz.to(dev), x.to(dev), dw, db, beta=0.0)
    assert err(dw, dz.double().t() @ x.double()) < 2e-3
    a = dw.clone()
    ops.linear_bwd_weight(dz.to(dev), x.to(dev), dw, db, beta=0.0)
    assert torch.equal(a, dw)    # deterministic combine


@pytest.mark.parametrize("M,N,K,tile", [(300, 136, 72, "64"), (1000, 512, 256, "128"), (700, 101, 64, "64")])
def test_live_row_gemms(dev, M, N, K, tile, monkeypatch):
    """row-set ("varlen") GEMMs: only listed slots are computed / written / reduced over"""
    ops, L = _ops()
    monkeypatch.setenv("NACF_GEMM_TILE", tile)
    g = torch.Generator().manual_seed(0)
    tok = torch.randint(0, 3, (M,), generator=g)                  # ~1/3 PAD slots, anywhere
    live = tok.ne(PAD)
    rs = ops.rowset_build(tokens=tok.to(dev))
    assert int(rs.count) == int(live.sum())
    assert torch.equal(rs.rows[:int(rs.count)].cpu().long(), live.nonzero().squeeze(1))
    flags = (torch.arange(M) % 2).to(torch.uint8)
    rs2 = ops.rowset_build(tokens=tok.to(dev), flags=flags.to(dev))
    assert torch.equal(rs2.rows[:int(rs2.count)].cpu().long(), (live & flags.bool()).nonzero().squeeze(1))
    x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3), rnd(M, N, seed=4)
    y = torch.full((M, N), 7.0, device=dev)
    pre = torch.full((M, N), 5.0, device=dev)
    ops.linear_fwd(x.to(dev), w.to(dev), y, ops.Epi(bias=b.to(dev), act=L.ACT_GELU_NEW, preact=pre, residual=r.to(dev),
                                                    row_tokens=tok.to(dev)), rows=rs)
    z = x.double() @ w.double().t() + b.double()
    ref = 0.5 * z * (1 + torch.tanh(math.sqrt(2 / math.pi) * (z + 0.044715 * z ** 3))) + r.double()
    assert err(y[live.to(dev)], ref[live]) < 1e-4 and err(pre[live.to(dev)], z[live]) < 1e-4
    assert float((y[~live.to(dev)] - 7.0).abs().max()) == 0 and float((pre[~live.to(dev)] - 5.0).abs().max()) == 0
    dz = rnd(M, N, seed=5)
    dx = torch.full((M, K), 3.0, device=dev)
    ops.linear_bwd_data(dz.to(dev), w.to(dev), dx, rows=rs)
    assert err(dx[live.to(dev)], (dz.double() @ w.double())[live]) < 1e-4
    assert float((dx[~live.to(dev)] - 3.0).abs().max()) == 0
    dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    ops.linear_bwd_weight(dz.to(dev), x.to(dev), dw, db, beta=0.0, rows=rs)
    assert err(dw, dz.double()[live].t() @ x.double()[live]) < 2e-4 * math.sqrt(M / 64)
    assert err(db, dz.double()[live].sum(0)) < 2e-4
    # empty list: nothing is touched, dW becomes zero
    none = ops.rowset_build(tokens=torch.zeros(M, dtype=torch.int64, device=dev))
    assert int(none.count) == 0
    ops.linear_fwd(x.to(dev), w.to(dev), y, None, rows=none)
    ops.linear_bwd_weight(dz.to(dev), x.to(dev), dw, None, beta=0.0, rows=none)
    assert float(dw.abs().max()) == 0


@pytest.mark.parametrize("M,N,K,tile", [(300, 136, 72, "64"), (1000, 512, 256, "128"), (130, 101, 64, "128"),
                                        (2560, 4100, 64, "64")])
def test_live_row_gemms_zero_fill_dead_rows(dev, M, N, K, tile, monkeypatch):
    """zero_dead: the workgroups of the dead row tiles write zeros, outputs need no memset (NaN-poisoned here)"""
    ops, L = _ops()
    monkeypatch.setenv("NACF_GEMM_TILE", tile)
    g = torch.Generator().manual_seed(1)
    tok = torch.randint(0, 3, (M,), generator=g)
    live = tok.ne(PAD)
    rs = ops.rowset_build(tokens=tok.to(dev))
    n_live = int(rs.count)
    assert torch.equal(rs.rows[:n_live].cpu().long(), live.nonzero().squeeze(1))
    assert torch.equal(rs.rows[n_live:].cpu().long().sort().values, (~live).nonzero().squeeze(1))   # the dead slots follow
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    y = torch.full((M, N), float("nan"), device=dev)
    pre = torch.full((M, N), float("nan"), device=dev)
    ops.linear_fwd(x.to(dev), w.to(dev), y, ops.Epi(bias=b.to(dev), act=L.ACT_RELU, preact=pre), rows=rs, zero_dead=True)
    z = x.double() @ w.double().t() + b.double()
    assert err(y[live.to(dev)], z.clamp_min(0)[live]) < 1e-4 and err(pre[live.to(dev)], z[live]) < 1e-4
    assert float(y[~live.to(dev)].abs().max()) == 0 and float(pre[~live.to(dev)].abs().max()) == 0
    dz = rnd(M, N, seed=5)
    dx = torch.full((M, K), float("nan"), device=dev)
    ops.linear_bwd_data(dz.to(dev), w.to(dev), dx, rows=rs, zero_dead=True)      # N = 4100: the split-K route
    assert err(dx[live.to(dev)], (dz.double() @ w.double())[live]) < 2e-4
    assert float(dx[~live.to(dev)].abs().max()) == 0
    # every slot dead / every slot live
    for toks in (torch.zeros(M, dtype=torch.int64), torch.ones(M, dtype=torch.int64)):
        rs2 = ops.rowset_build(tokens=toks.to(dev))
        y.fill_(float("nan"))
        ops.linear_fwd(x.to(dev), w.to(dev), y, ops.Epi(bias=b.to(dev)), rows=rs2, zero_dead=True)
        want = z if int(toks[0]) else torch.zeros_like(z)
        assert err(y, want) < 1e-4


def test_vocab_argmax_live_rows(dev):
    ops, _ = _ops()
    rows, V, K = 200, 333, 64
    h, w = rnd(rows, K, seed=1), rnd(V, K, seed=2, scale=0.5)
    pad = torch.randint(0, 3, (rows,), generator=torch.Generator().manual_seed(3))
    upd = (torch.arange(rows) % 3 == 0).to(torch.uint8)
    full_t = torch.full((rows,), 9, dtype=torch.int64, device=dev)
    full_p = torch.full((rows,), 0.5, device=dev)
    ops.vocab_argmax(h.to(dev), w.to(dev), None, pad.to(dev), False, upd.to(dev), full_t, full_p)
    rs = ops.rowset_build(tokens=pad.to(dev), flags=upd.to(dev))
    t2 = torch.full((rows,), 9, dtype=torch.int64, device=dev)
    p2 = torch.full((rows,), 0.5, device=dev)
    ops.vocab_argmax(h.to(dev), w.to(dev), None, pad.to(dev), False, upd.to(dev), t2, p2, rows=rs)
    sel = (pad.ne(0) & upd.bool()).to(dev)
    assert torch.equal(t2[sel], full_t[sel]) and torch.equal(p2[sel], full_p[sel])   # same math on the live slots
    assert torch.equal(t2[~sel], torch.full_like(t2[~sel], 9))                        # everything else untouched


def test_epilogue_bwd_matches_autograd(dev):
    ops, L = _ops()
    M, N = 64, 128
    z = rnd(M, N, seed=1).double().requires_grad_(True)
    r = rnd(M, N, seed=2).double().requires_grad_(True)
    tok = torch.randint(0, 2, (M,), generator=torch.Generator().manual_seed(3))
    y = (0.5 * z * (1 + torch.tanh(math.sqrt(2 / math.pi) * (z + 0.044715 * z ** 3))) + r) * tok.ne(0).double().unsqueeze(1)
    dy = rnd(M, N, seed=4)
    y.backward(dy.double())
    epi = ops.Epi(act=L.ACT_GELU_NEW, preact=z.detach().float().to(dev), residual=r.detach().float().to(dev),
                  row_tokens=tok.to(dev))
    dz = torch.empty(M, N, device=dev)
    dr = torch.empty(M, N, device=dev)
    ops.epilogue_bwd(dy.to(dev), dz, dr, epi)
    assert err(dz, z.grad) < 1e-5 and err(dr, r.grad) < 1e-6


@pytest.mark.parametrize("N", [256, 250])      # 250: not a multiple of 4 -> the scalar kernel (same row handling: ADVICE round 4)
def test_epilogue_bwd_with_a_live_row_list(dev, N):
    """the forward GEMM's row list: live rows are bit-identical to the all-rows launch (same dropout masks: functions of the
    physical element index), dead rows of dZ are left untouched, dead rows of dR get zeros without anything being read
    (dY / the pre-activation hold NaNs there), and with accumulate_dR they are left alone"""
    ops, L = _ops()
    M = 300
    tok = torch.randint(0, 3, (M,), generator=torch.Generator().manual_seed(3))
    live = tok.ne(0)
    rows = ops.rowset_build(tokens=tok.to(dev))
    rng = ops.RngState(99, dev)
    z = rnd(M, N, seed=1)
    dy = rnd(M, N, seed=4)
    mk = lambda pre: ops.Epi(act=L.ACT_GELU_NEW, preact=pre.to(dev), p1=0.3, salt1=7, residual=torch.empty(M, N, device=dev),
                             p2=0.2, salt2=8, row_tokens=tok.to(dev), rng=rng)
    dz0, dr0 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    ops.epilogue_bwd(dy.to(dev), dz0, dr0, mk(z))
    # poison what must not be read
    z_p, dy_p = z.clone(), dy.clone()
    z_p[~live] = float("nan")
    dy_p[~live] = float("nan")
    dz1, dr1 = torch.full((M, N), 7.0, device=dev), torch.full((M, N), 7.0, device=dev)
    ops.epilogue_bwd(dy_p.to(dev), dz1, dr1, mk(z_p), rows=rows)
    lv = live.to(dev)
    assert torch.equal(dz1[lv], dz0[lv]) and torch.equal(dr1[lv], dr0[lv]) and bool(dz1[lv].ne(0).any())
    assert float((dz1[~lv] - 7.0).abs().max()) == 0.0                  # untouched
    assert float(dr1[~lv].abs().max()) == 0.0 and float(dr0[~lv].abs().max()) == 0.0
    acc = torch.full((M, N), 2.0, device=dev)
    ops.epilogue_bwd(dy_p.to(dev), dz1, acc, mk(z_p), accumulate_dr=True, rows=rows)
    assert torch.equal(acc[lv], dr0[lv] + 2.0) and float((acc[~lv] - 2.0).abs().max()) == 0.0


@pytest.mark.parametrize("rows,V,K", [(57, 101, 64), (300, 1000, 64), (128, 10547, 512)])
def test_vocab_argmax_fused(dev, rows, V, K):
    ops, _ = _ops()
    h, w = rnd(rows, K, seed=1), rnd(V, K, seed=2, scale=0.5)
    logits = h.double() @ w.double().t()
    probs = torch.softmax(logits, -1)
    rp, ri = probs.max(-1)
    pad = torch.randint(0, 4, (rows,), generator=torch.Generator().manual_seed(3))
    tok = torch.full((rows,), -1, dtype=torch.int64, device=dev)
    pr = torch.full((rows,), -1.0, device=dev)
    ops.vocab_argmax(h.to(dev), w.to(dev), None, pad.to(dev), False, None, tok, pr)
    exp_tok = torch.where(pad.eq(0), torch.zeros_like(ri), ri)
    exp_p = torch.where(pad.eq(0), torch.ones_like(rp), rp)
    top2 = logits.topk(2, -1)[0]
    safe = ((top2[:, 0] - top2[:, 1]) > 1e-4) | pad.eq(0)
    assert torch.equal(tok.cpu()[safe], exp_tok[safe])
    assert err(pr.cpu()[safe], exp_p[safe]) < 1e-5
    # update mask: only flagged rows change; zero_mask_prob zeroes prob where the argmax is <mask>
    upd = (torch.arange(rows) % 2).to(torch.uint8)
    tok2 = torch.full((rows,), 7, dtype=torch.int64, device=dev)
    pr2 = torch.full((rows,), 0.25, device=dev)
    ops.vocab_argmax(h.to(dev), w.to(dev), None, pad.to(dev), False, upd.to(dev), tok2, pr2)
    assert torch.equal(tok2.cpu()[upd == 0], torch.full_like(tok2.cpu()[upd == 0], 7))
    assert torch.equal(tok2.cpu()[(upd == 1) & safe], exp_tok[(upd == 1) & safe])
    w2 = w.clone(); w2[MASK] = h[0] * 50          # row 0 is forced to predict <mask>
    pad0 = pad.clone(); pad0[0] = 9
    ops.vocab_argmax(h.to(dev), w2.to(dev), None, pad0.to(dev), True, None, tok, pr)
    assert int(tok[0]) == MASK and float(pr[0]) == 0.0


# ------------------------------------------------------------------ encoder tail
def test_highway_mix_fwd_bwd(dev):
    ops, _ = _ops()
    rows, D = 100, 64
    h = rnd(rows, D, seed=1).double().requires_grad_(True)
    p1 = rnd(rows, D, seed=2).double().requires_grad_(True)
    p2 = rnd(rows, D, seed=3).double().requires_grad_(True)
    t_, g_ = torch.tanh(p1), torch.sigmoid(p2)
    out = g_ * h + (1 - g_) * t_
    do = rnd(rows, D, seed=4)
    out.backward(do.double())
    tg = torch.cat([t_, g_], 1).detach().float().to(dev)
    o = torch.empty(rows, D, device=dev)
    ops.highway_mix_fwd(h.detach().float().to(dev), tg, o, 0.0, 0, None)
    assert err(o, out) < 1e-6
    dh = torch.empty(rows, D, device=dev)
    dp = torch.empty(rows, 2 * D, device=dev)
    ops.highway_mix_bwd(do.to(dev), h.detach().float().to(dev), tg, dh, dp, 0.0, 0, None)
    assert err(dh, h.grad) < 1e-6            # direct path only (h also feeds p1/p2 through the GEMM)
    assert err(dp[:, :D], p1.grad) < 1e-6 and err(dp[:, D:], p2.grad) < 1e-6


@pytest.mark.parametrize("B,Fr,D", [(3, 6, 64), (16, 60, 512), (5, 7, 100)])
def test_bn_concat_fwd_bwd(dev, B, Fr, D):
    ops, _ = _ops()
    x = (rnd(B, Fr, D, seed=1) * 2 + 0.5)
    w, b = rnd(D, seed=2) + 1.5, rnd(D, seed=3)
    bn = torch.nn.BatchNorm1d(D).double()
    with torch.no_grad():
        bn.weight.copy_(w); bn.bias.copy_(b)
    xr = x.double().requires_grad_(True)
    yr = bn(xr.view(B * Fr, D)).view(B, Fr, D)
    M_total, f_off = Fr + 5, 3
    dout = rnd(B, M_total, D, seed=4)
    yr.backward(dout[:, f_off:f_off + Fr].double())
    out = torch.zeros(B, M_total, D, device=dev)
    rm, rv = torch.zeros(D, device=dev), torch.ones(D, device=dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    sm, si = torch.empty(D, device=dev), torch.empty(D, device=dev)
    xd = x.to(dev)
    ops.bn_concat_fwd(xd, out, f_off, w.to(dev), b.to(dev), rm, rv, nbt, sm, si, True)
    assert err(out[:, f_off:f_off + Fr], yr) < 2e-5
    assert float(out[:, :f_off].abs().max()) == 0
    assert err(rm, bn.running_mean) < 1e-6 and err(rv, bn.running_var) < 1e-5 and int(nbt) == 1
    dx = torch.empty_like(xd)
    dw, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    ops.bn_concat_bwd(dout.to(dev), xd, dx, f_off, w.to(dev), sm, si, dw, db, beta=1.0)
    assert err(dx, xr.grad) < 5e-5
    assert err(dw, bn.weight.grad) < 2e-4 and err(db, bn.bias.grad) < 2e-4
    # eval mode uses the running statistics
    bn.eval()
    out2 = torch.zeros(B, M_total, D, device=dev)
    ops.bn_concat_fwd(xd, out2, f_off, w.to(dev), b.to(dev), rm, rv, nbt, None, None, False)
    assert err(out2[:, f_off:f_off + Fr], bn(x.double().view(B * Fr, D)).view(B, Fr, D)) < 2e-5


def test_mean_time_logsoftmax_kldiv(dev):
    ops, _ = _ops()
    x = rnd(7, 13, 96, seed=1)
    out = torch.empty(7, 96, device=dev)
    ops.mean_time_fwd(x.to(dev), out)
    assert err(out, x.double().mean(1)) < 1e-6
    dx = torch.empty(7, 13, 96, device=dev)
    ops.mean_time_bwd(out, dx)
    assert err(dx, (out.double().cpu() / 13).unsqueeze(1).expand(-1, 13, -1)) < 1e-7
    # two upstream gradients of the mean (its two consumers) added inside, accumulate on top of dx; float4 and scalar forms
    for D in (96, 30):
        g1, g2, base = rnd(7, D, seed=5).to(dev), rnd(7, D, seed=6).to(dev), rnd(7, 13, D, seed=7).to(dev)
        d2 = base.clone()
        ops.mean_time_bwd(g1, d2, accumulate=True, dout2=g2)
        want = base.double().cpu() + ((g1.double().cpu() + g2.double().cpu()) / 13).unsqueeze(1)
        assert err(d2, want) < 1e-6
    z = rnd(9, 30, seed=2, scale=3).double().requires_grad_(True)
    lp = torch.log_softmax(z, -1)
    tgt = torch.zeros(9, 30, dtype=torch.float64); tgt[torch.arange(9), torch.arange(9) + 4] = 1.0
    loss = F.kl_div(lp, tgt, reduction="mean")
    loss.backward()
    zd = z.detach().float().to(dev)
    lpd = torch.empty_like(zd)
    ops.log_softmax_rows(zd, lpd)
    assert err(lpd, lp) < 1e-5
    lo = torch.empty(1, device=dev)
    ops.kldiv_mean(lpd, tgt.float().to(dev), lo, None)
    assert abs(float(lo) - float(loss)) < 1e-6
    dlp = torch.empty_like(lpd)
    ops.kldiv_mean(lpd, tgt.float().to(dev), None, dlp, gscale=torch.ones(1, device=dev))
    dz = torch.empty_like(lpd)
    ops.log_softmax_rows_bwd(dlp, lpd, dz)
    assert err(dz, z.grad) < 1e-6


# ------------------------------------------------------------------ decoder
@pytest.mark.parametrize("R,Lq,D,V,with_cat", [(6, 10, 64, 101, True), (4, 7, 512, 300, False)])
def test_embed_ln_fwd_bwd_scatter(dev, R, Lq, D, V, with_cat):
    ops, _ = _ops()
    Bv = R // 2
    g = torch.Generator().manual_seed(0)
    tok = torch.randint(0, 8, (R, Lq), generator=g)        # few distinct ids -> heavy collisions, PAD present
    cat_ids = torch.randint(0, 5, (Bv,), generator=g)
    word = rnd(V, D, seed=1).double().requires_grad_(True)
    pos = rnd(Lq, D, seed=2).double().requires_grad_(True)
    cat = rnd(5, D, seed=3).double().requires_grad_(True)
    add = rnd(Bv, D, seed=4).double().requires_grad_(True)
    lw = (rnd(D, seed=5) + 1.5).double().requires_grad_(True)
    lb = rnd(D, seed=6).double().requires_grad_(True)
    vid = torch.arange(R) % Bv                              # pass-major map (vdiv=1, vmod=Bv)
    e = word[tok] + pos[:Lq].unsqueeze(0)
    if with_cat:
        e = e + cat[cat_ids[vid]].unsqueeze(1)
    e = e + add[vid].unsqueeze(1)
    y = F.layer_norm(e, (D,), lw, lb, 1e-5)
    dy = rnd(R, Lq, D, seed=7)
    y.backward(dy.double())
    f = lambda t_: t_.detach().float().to(dev)
    out = torch.empty(R, Lq, D, device=dev)
    xhat = torch.empty(R * Lq, D, device=dev)
    rstd = torch.empty(R * Lq, device=dev)
    ops.embed_ln_fwd(tok.to(dev), cat_ids.to(dev) if with_cat else None, f(add), f(word), f(pos),
                     f(cat) if with_cat else None, f(lw), f(lb), out, xhat, rstd, 1, Bv, 1e-5, 0.0, 0, None)
    assert err(out, y) < 2e-5
    dE = torch.empty(R * Lq, D, device=dev)
    dlw, dlb = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    ops.embed_ln_bwd(dy.to(dev), xhat, rstd, f(lw), dE, dlw, dlb, R, Lq, D, 0.0, 0, None, beta=1.0)
    assert err(dlw, lw.grad) < 2e-4 and err(dlb, lb.grad) < 2e-4
    dword, dpos = torch.zeros(V, D, device=dev), torch.zeros(Lq, D, device=dev)
    dcat = torch.zeros(5, D, device=dev)
    dadd = torch.empty(Bv, D, device=dev)
    ops.embed_scatter_bwd(dE, tok.to(dev), cat_ids.to(dev) if with_cat else None, dword, dpos,
                          dcat if with_cat else None, dadd, R, Lq, D, V, 5, Bv, 1, Bv)
    wg = word.grad.clone(); wg[PAD] = 0                     # padding_idx row gets no gradient
    assert err(dword, wg) < 2e-4
    assert err(dpos, pos.grad) < 2e-4 and err(dadd, add.grad) < 2e-4
    if with_cat:
        assert err(dcat, cat.grad) < 2e-4
    d2 = torch.zeros(V, D, device=dev)
    ops.embed_scatter_bwd(dE, tok.to(dev), None, d2, None, None, None, R, Lq, D, V, 0, Bv, 1, Bv)
    assert torch.equal(d2, dword)                           # fixed summation order


def _mha_ref(q, k, v, H, key_pad, causal):
    R, Lq, D = q.shape
    Lk = k.shape[1]
    dk = D // H
    qh = q.view(R, Lq, H, dk).permute(0, 2, 1, 3)
    kh = k.view(R, Lk, H, dk).permute(0, 2, 1, 3)
    vh = v.view(R, Lk, H, dk).permute(0, 2, 1, 3)
    s = qh @ kh.transpose(-1, -2) / math.sqrt(dk)
    m = torch.zeros(R, 1, Lq, Lk, dtype=torch.bool)
    if key_pad is not None:
        m = m | key_pad.view(R, 1, 1, Lk)
    if causal:
        m = m | torch.triu(torch.ones(Lq, Lk, dtype=torch.bool), 1).view(1, 1, Lq, Lk)
    s = s.masked_fill(m, -10e6)
    p = torch.softmax(s, -1)
    return (p @ vh).permute(0, 2, 1, 3).reshape(R, Lq, D), p.permute(1, 0, 2, 3)


@pytest.mark.parametrize("R,Lq,H,dk,causal", [(5, 7, 4, 16, False), (3, 9, 4, 16, True), (4, 20, 8, 64, False)])
def test_self_attention_fwd_bwd(dev, R, Lq, H, dk, causal):
    ops, _ = _ops()
    D = H * dk
    qkv = rnd(R * Lq, 3 * D, seed=1)
    tok = torch.randint(1, 9, (R, Lq), generator=torch.Generator().manual_seed(2))
    tok[0, Lq - 3:] = PAD
    tok[1, 2] = PAD                                           # PAD in the middle (decode can predict id 0)
    qd = qkv.double().requires_grad_(True)
    q, k, v = [qd[:, i * D:(i + 1) * D].reshape(R, Lq, D) for i in range(3)]
    o_ref, p_ref = _mha_ref(q, k, v, H, tok.eq(PAD), causal)
    do = rnd(R * Lq, D, seed=3)
    o_ref.backward(do.view(R, Lq, D).double())
    x = qkv.to(dev)
    out = torch.empty(R * Lq, D, device=dev)
    probs = torch.empty(H, R, Lq, Lq, device=dev)
    ops.attention_fwd(x[:, :D], x[:, D:2 * D], x[:, 2 * D:], out, tok.to(dev), causal, probs, R, H, Lq, Lq, dk, 1, R)
    assert err(out, o_ref.reshape(R * Lq, D)) < 2e-5
    assert err(probs, p_ref) < 1e-5
    dqkv = torch.empty_like(x)
    ops.attention_bwd(x[:, :D], x[:, D:2 * D], x[:, 2 * D:], do.to(dev), dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:],
                      tok.to(dev), causal, R, R, H, Lq, Lq, dk, 1, R)
    assert err(dqkv, qd.grad) < 5e-5


@pytest.mark.parametrize("wpi", ["1", "2"])
@pytest.mark.parametrize("mode", ["mod", "div"])
def test_cross_attention_shared_memory_fwd_bwd(dev, mode, wpi, monkeypatch):
    monkeypatch.setenv("NACF_ATTN_WPI", wpi)      # waves per memory row set in the backward kernel (default 1)
    ops, _ = _ops()
    Bv, k_, Lq, Lk, H, dk = 3, 2, 6, 12, 4, 16
    D, R = H * dk, Bv * k_
    q = rnd(R * Lq, D, seed=1)
    kv = rnd(Bv * Lk, 2 * D, seed=2)
    vid = (torch.arange(R) % Bv) if mode == "mod" else (torch.arange(R) // k_)
    kv_div, kv_mod = (1, Bv) if mode == "mod" else (k_, Bv)
    qd = q.double().requires_grad_(True)
    kvd = kv.double().requires_grad_(True)
    kk = kvd[:, :D].reshape(Bv, Lk, D)[vid]
    vv = kvd[:, D:].reshape(Bv, Lk, D)[vid]
    o_ref, _ = _mha_ref(qd.view(R, Lq, D), kk, vv, H, None, False)
    do = rnd(R * Lq, D, seed=3)
    o_ref.backward(do.view(R, Lq, D).double())
    qx, kvx = q.to(dev), kv.to(dev)
    out = torch.empty(R * Lq, D, device=dev)
    ops.attention_fwd(qx, kvx[:, :D], kvx[:, D:], out, None, 0, None, R, H, Lq, Lk, dk, kv_div, kv_mod)
    assert err(out, o_ref.reshape(R * Lq, D)) < 2e-5
    dq, dkv = torch.empty_like(qx), torch.empty_like(kvx)
    ops.attention_bwd(qx, kvx[:, :D], kvx[:, D:], do.to(dev), dq, dkv[:, :D], dkv[:, D:], None, 0, R, Bv, H, Lq, Lk, dk,
                      kv_div, kv_mod)
    assert err(dq, qd.grad) < 5e-5 and err(dkv, kvd.grad) < 5e-5   # memory grads sum over the rows sharing a video


def test_attention_full_size_cross(dev):
    ops, _ = _ops()
    Bv, lbs, Lq, Lk, H, dk = 2, 3, 19, 120, 8, 64
    D, R = 512, Bv * lbs
    q, kv = rnd(R * Lq, D, seed=1), rnd(Bv * Lk, 2 * D, seed=2)
    vid = torch.arange(R) // lbs
    o_ref, _ = _mha_ref(q.double().view(R, Lq, D), kv[:, :D].double().view(Bv, Lk, D)[vid],
                        kv[:, D:].double().view(Bv, Lk, D)[vid], H, None, False)
    out = torch.empty(R * Lq, D, device=dev)
    kvx = kv.to(dev)
    ops.attention_fwd(q.to(dev), kvx[:, :D], kvx[:, D:], out, None, 0, None, R, H, Lq, Lk, dk, lbs, Bv)
    assert err(out, o_ref.reshape(R * Lq, D)) < 5e-5
    # the candidates of a video as ONE sequence of lbs*Lq queries (blocks of 32 queries per wave): same bits
    out2 = torch.empty(R * Lq, D, device=dev)
    ops.attention_fwd(q.to(dev), kvx[:, :D], kvx[:, D:], out2, None, 0, None, Bv, H, lbs * Lq, Lk, dk, 1, Bv)
    assert torch.equal(out, out2)
    probs = torch.empty(H, Bv, lbs * Lq, Lk, device=dev)
    ops.attention_fwd(q.to(dev), kvx[:, :D], kvx[:, D:], out2, None, 0, probs, Bv, H, lbs * Lq, Lk, dk, 1, Bv)
    assert torch.equal(out, out2) and float((probs.sum(-1) - 1).abs().max()) < 1e-5


@pytest.mark.parametrize("R,Lq,H,dk,causal", [(5, 7, 4, 16, False), (3, 9, 4, 16, True), (2, 20, 8, 64, False)])
def test_attention_generic_lds_kernels(dev, R, Lq, H, dk, causal, monkeypatch):
    """the LDS/VALU attention kernels (shapes the MFMA path does not cover) stay correct"""
    monkeypatch.setenv("NACF_ATTN_VALU", "1")
    test_self_attention_fwd_bwd(dev, R, Lq, H, dk, causal)
    test_cross_attention_shared_memory_fwd_bwd(dev, "mod", "1", monkeypatch)
    test_cross_attention_shared_memory_fwd_bwd(dev, "div", "1", monkeypatch)


def test_attention_odd_head_dim_uses_generic_path(dev):
    ops, _ = _ops()
    R, Lq, H, dk = 3, 6, 2, 24                      # dk = 24: not an MFMA-path shape
    D = H * dk
    qkv = rnd(R * Lq, 3 * D, seed=1)
    tok = torch.randint(1, 9, (R, Lq), generator=torch.Generator().manual_seed(2))
    q, k, v = [qkv[:, j * D:(j + 1) * D].double().reshape(R, Lq, D) for j in range(3)]
    o_ref, _ = _mha_ref(q, k, v, H, tok.eq(PAD), False)
    x = qkv.to(dev)
    out = torch.empty(R * Lq, D, device=dev)
    ops.attention_fwd(x[:, :D], x[:, D:2 * D], x[:, 2 * D:], out, tok.to(dev), False, None, R, H, Lq, Lq, dk, 1, R)
    assert err(out, o_ref.reshape(R * Lq, D)) < 2e-5


def test_masked_mean(dev):
    ops, _ = _ops()
    y = rnd(4, 6, 32, seed=1)
    tok = torch.tensor([[5, 6, 7, 0, 0, 0], [5, 5, 5, 5, 5, 5], [9, 0, 0, 0, 0, 0], [3, 4, 0, 0, 0, 0]])
    out = torch.empty(4, 32, device=dev)
    ops.masked_mean_fwd(y.to(dev), tok.to(dev), out)
    assert err(out, y.double().sum(1) / tok.ne(0).sum(1, keepdim=True).double()) < 1e-6


# ------------------------------------------------------------------ vocabulary / loss
@pytest.mark.parametrize("rows,V", [(33, 101), (64, 10547)])
def test_vocab_logsoftmax_nll_xent(dev, rows, V):
    ops, _ = _ops()
    ld = ops.vocab_ld(V)
    z = rnd(rows, V, seed=1, scale=4)
    labels = torch.randint(0, 9, (rows,), generator=torch.Generator().manual_seed(2))
    zr = z.double().requires_grad_(True)
    lp_ref = torch.log_softmax(zr, -1)
    m = labels.ne(PAD)
    loss = -(lp_ref.gather(1, labels.view(-1, 1)).squeeze(1) * m).sum()
    (loss * 0.7).backward()
    buf = torch.zeros(rows, ld, device=dev)
    buf[:, :V] = z.to(dev)
    lp = buf[:, :V]
    lse = torch.empty(rows, device=dev)
    am = torch.empty(rows, dtype=torch.int64, device=dev)
    ll = torch.empty(rows, device=dev)
    ops.vocab_logsoftmax_fwd(lp, V, labels.to(dev), lse, am, ll)
    assert err(lp, lp_ref) < 2e-5 and err(lse, torch.logsumexp(z.double(), -1)) < 2e-5
    assert torch.equal(am.cpu(), z.argmax(-1))
    out5 = torch.empty(5, device=dev)
    ops.nll_reduce(ll, am, labels.to(dev), True, out5)
    ind = m & labels.ne(MASK)
    exp = [float(loss), float((z.argmax(-1).eq(labels) & ind).sum()), float(ind.sum()), -float(loss), float(m.sum())]
    assert all(abs(a - b) < 1e-3 * max(1, abs(b)) for a, b in zip(out5.tolist(), exp)), (out5.tolist(), exp)
    g = torch.tensor([0.7], device=dev)
    ops.xent_bwd(lp, lp, V, labels.to(dev), g, 1.0)          # in place
    assert err(lp, zr.grad) < 1e-5
    assert float(buf[:, V:].abs().max()) == 0
    # generic log-softmax backward
    dlp = rnd(rows, V, seed=3)
    lp2 = torch.log_softmax(zr.detach(), -1).float().to(dev)
    dz = torch.empty(rows, V, device=dev)
    ops.vocab_logsoftmax_bwd(dlp.to(dev), lp2, dz, V)
    ref = dlp.double() - torch.softmax(z.double(), -1) * dlp.double().sum(-1, keepdim=True)
    assert err(dz, ref) < 2e-5


def test_passes_back_to_back_in_one_launch(dev):
    """nacf_nll_reduce_multi / nacf_xent_bwd_lse_multi: the decoding passes of one batch (rows back to back) in one launch
    each -- bit-identical to one nacf_nll_reduce / nacf_xent_bwd_lse call per pass (own exclusion rule, own upstream gradient)"""
    ops, _ = _ops()
    rp, n_pass, V = 77, 3, 530
    rows = rp * n_pass
    ld = ops.vocab_ld(V)
    buf = torch.zeros(rows, ld, device=dev)
    buf[:, :V] = rnd(rows, V, seed=1, scale=4).to(dev)
    labels = torch.randint(0, 9, (rows,), generator=torch.Generator().manual_seed(2)).to(dev)
    logits = buf[:, :V]
    lse = torch.logsumexp(logits.double(), -1).float()
    am = logits.argmax(-1)
    ll = (logits.gather(1, labels.view(-1, 1)).squeeze(1) - lse).contiguous()
    excl = [True, False, True]
    one = [torch.empty(5, device=dev) for _ in range(n_pass)]
    for i in range(n_pass):
        sl = slice(i * rp, (i + 1) * rp)
        ops.nll_reduce(ll[sl], am[sl], labels[sl], excl[i], one[i])
    many = [torch.empty(5, device=dev) for _ in range(n_pass)]
    ops.nll_reduce_multi(ll, am, labels, excl, many)
    assert all(torch.equal(a, b) for a, b in zip(one, many))
    gs = [torch.tensor([0.7, 0, 0, 0, 0], device=dev), torch.tensor([-1.3, 0, 0, 0, 0], device=dev), torch.tensor([0.0, 0, 0, 0, 0], device=dev)]
    d1 = torch.full((rows, ld), 7.0, device=dev)
    for i in range(n_pass):
        sl = slice(i * rp, (i + 1) * rp)
        ops.xent_bwd_lse(logits[sl], lse[sl], d1[sl, :V], V, labels[sl], gs[i], 1.0, skip_pad_rows=True)
    d2 = torch.full((rows, ld), 7.0, device=dev)
    ops.xent_bwd_lse_multi(logits, lse, d2[:, :V], V, labels, gs, 1.0, skip_pad_rows=True)
    assert torch.equal(d1, d2)
    live = labels.ne(PAD).cpu()
    ref = (torch.softmax(logits.double().cpu(), -1) - torch.nn.functional.one_hot(labels.cpu(), V).double())
    scale = torch.tensor([0.7] * rp + [-1.3] * rp + [0.0] * rp, dtype=torch.float64).view(-1, 1)
    assert err(d2[:, :V][live.to(dev)], (ref * scale)[live]) < 1e-5 and float(d2[~live.to(dev)].sub(7.0).abs().max()) == 0


# ------------------------------------------------------------------ decode bookkeeping (bit-exact)
def test_length_beam_and_canvas(dev):
    ops, _ = _ops()
    B, max_len, lbs = 9, 20, 6
    pl = torch.log_softmax(rnd(B, max_len, seed=1, scale=3), -1)
    beam = torch.empty(B, lbs, dtype=torch.int32, device=dev)
    bmax = torch.empty(1, dtype=torch.int32, device=dev)
    ops.length_beam(pl.to(dev), lbs, 0, beam, bmax)
    ref = pl.topk(lbs, dim=1)[1].clamp(4, max_len - 1)
    assert torch.equal(beam.cpu().long(), ref) and int(bmax) == int(ref.max())
    Lp = int(bmax)
    tok = torch.empty(B * lbs, Lp, dtype=torch.int64, device=dev)
    ops.canvas_init(beam, B * lbs, Lp, tok)
    exp = torch.where(torch.arange(Lp).view(1, -1) < ref.view(-1, 1), MASK, PAD)
    assert torch.equal(tok.cpu(), exp)


def test_select_mask_matches_topk(dev):
    ops, _ = _ops()
    rows, Lp, T = 40, 19, 6
    g = torch.Generator().manual_seed(0)
    lens = torch.randint(4, Lp + 1, (rows,), generator=g)
    pad_tokens = torch.where(torch.arange(Lp).view(1, -1) < lens.view(-1, 1), MASK, PAD)
    probs = torch.rand(rows, Lp, generator=g)
    probs[pad_tokens.eq(PAD)] = 1.0
    teacher = torch.rand(rows, Lp, generator=g)
    for c in range(1, T):
        ratio = 1.0 - c / T
        lut = (torch.arange(Lp + 1).float() * ratio).long().to(torch.int32)
        num = (lens.float() * ratio).long()
        exp = torch.zeros(rows, Lp, dtype=torch.bool)
        sc = probs * teacher
        for i in range(rows):
            exp[i, sc[i].topk(max(1, int(num[i])), largest=False)[1]] = True
        tok = torch.full((rows, Lp), 9, dtype=torch.int64, device=dev)
        mask = torch.empty(rows, Lp, dtype=torch.uint8, device=dev)
        ops.select_mask(probs.to(dev), teacher.to(dev), pad_tokens.to(dev), lut.to(dev), 0, tok, mask)
        assert torch.equal(mask.cpu().bool(), exp)
        assert torch.equal(tok.cpu().eq(MASK), exp)
    tok = torch.where(torch.rand(rows, Lp, generator=g) < 0.3, MASK, 9).to(dev)
    mask = torch.empty(rows, Lp, dtype=torch.uint8, device=dev)
    ops.select_mask(None, None, pad_tokens.to(dev), None, 1, tok, mask)
    assert torch.equal(mask.cpu().bool(), tok.cpu().eq(MASK))
    ops.select_mask(None, None, pad_tokens.to(dev), None, 2, tok.clone(), mask)
    assert torch.equal(mask.cpu().bool(), tok.cpu().ne(MASK) & pad_tokens.ne(PAD))


def test_best_candidate(dev):
    ops, _ = _ops()
    B, lbs, Lp, alpha = 7, 6, 15, 1.35
    g = torch.Generator().manual_seed(0)
    beam = torch.randint(4, Lp + 1, (B, lbs), generator=g)
    tokens = torch.randint(6, 99, (B * lbs, Lp), generator=g)
    probs = torch.rand(B * lbs, Lp, generator=g) * 0.9 + 0.05
    pad = torch.arange(Lp).view(1, -1) >= beam.view(-1, 1)
    probs[pad] = 1.0
    tokens[pad] = PAD
    lp = probs.log().view(B, lbs, Lp)
    best = (lp.sum(-1) / beam.float() ** alpha).max(-1)[1]
    exp = tokens.view(B, lbs, Lp).gather(1, best.view(B, 1, 1).expand(B, 1, Lp)).squeeze(1)
    out = torch.empty(B, Lp, dtype=torch.int64, device=dev)
    bi = torch.empty(B, dtype=torch.int32, device=dev)
    cl = torch.empty(B * lbs, Lp, device=dev)
    ops.best_candidate(tokens.to(dev), probs.to(dev), None, beam.to(torch.int32).to(dev), alpha, B, lbs, Lp, out, bi, cl)
    assert torch.equal(bi.cpu().long(), best) and torch.equal(out.cpu(), exp)
    assert err(cl, lp.view(B * lbs, Lp)) < 1e-6


def test_l2r_ef_helpers(dev):
    ops, _ = _ops()
    rows, Lp, q = 12, 11, 2
    g = torch.Generator().manual_seed(0)
    tok = torch.where(torch.rand(rows, Lp, generator=g) < 0.5, MASK, 9)
    tok[0] = 9
    rank = torch.empty(rows, Lp, dtype=torch.int32, device=dev)
    counts = torch.empty(2, dtype=torch.int32, device=dev)
    ops.mask_rank(tok.to(dev), rank, counts)
    m = tok.eq(MASK)
    exp_rank = torch.where(m, m.long().cumsum(1) - 1, torch.full_like(tok, -1))
    assert torch.equal(rank.cpu().long(), exp_rank)
    assert counts.tolist() == [int(m.sum(1).max()), int(m.sum())]
    t2 = torch.full((rows, Lp), 9, dtype=torch.int64, device=dev)
    mk = torch.empty(rows, Lp, dtype=torch.uint8, device=dev)
    ops.select_rank(rank, 2, q, t2, mk)
    assert torch.equal(mk.cpu().bool(), (exp_rank >= 2) & (exp_rank < 4))
    new_tok = torch.randint(6, 50, (rows, Lp), generator=g)
    new_p = torch.rand(rows, Lp, generator=g)
    probs = torch.zeros(rows, Lp)
    exp_t, exp_p = tok.clone(), probs.clone()
    for i in range(rows):
        r = int(m[i].sum())
        if r:
            cand = new_p[i].masked_fill(~m[i], 0)
            ind = cand.topk(min(q, r))[1]
            exp_t[i, ind] = new_tok[i, ind]; exp_p[i, ind] = cand[ind]
    td, pd = tok.to(dev), probs.to(dev)
    ops.easy_first_update(td, pd, new_tok.to(dev), new_p.to(dev), q)
    assert torch.equal(td.cpu(), exp_t) and torch.equal(pd.cpu(), exp_p)
    ops.token_replace(td, MASK, VIS)
    assert torch.equal(td.cpu(), torch.where(exp_t.eq(MASK), VIS, exp_t))
    pr = torch.empty(rows, Lp, device=dev)
    ops.init_probs(tok.masked_fill(tok.eq(9), PAD).to(dev), pr)
    assert torch.equal(pr.cpu(), tok.eq(9).float())


# ------------------------------------------------------------------ optimiser
def test_adam_step_matches_torch(dev):
    ops, _ = _ops()
    n = 10007
    p0, g0 = rnd(n, seed=1), rnd(n, seed=2, scale=8)          # some |g| > 5 -> clip path
    p = torch.nn.Parameter(p0.double().clone())
    opt = torch.optim.Adam([p], lr=5e-4, weight_decay=5e-4)
    pd = p0.to(dev).clone()
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    step = torch.zeros(1, dtype=torch.int64, device=dev)
    lr = torch.tensor([5e-4], device=dev)
    for it in range(3):
        g = g0 * (it + 1) / 2
        p.grad = (g.double() * 0.5).clamp(-5, 5)              # grad_scale 0.5, then clip
        opt.step()
        ops.adam_step(pd, g.to(dev), m, v, lr, step, 0.9, 0.999, 1e-8, 5e-4, 5.0, 0.5)
    assert int(step) == 3
    assert err(pd, p) < 2e-6


# ------------------------------------------------------------------ generic LayerNorm (with_layernorm / norm_type=ln)
@pytest.mark.parametrize("B,Fr,D", [(3, 6, 64), (5, 8, 512), (2, 7, 100)])
def test_layernorm_segment_write_and_mask(dev, B, Fr, D):
    ops, _ = _ops()
    x = rnd(B * Fr, D, seed=1) * 2 + 0.3
    w, b = rnd(D, seed=2) + 1.5, rnd(D, seed=3)
    toks = torch.randint(0, 3, (B * Fr,), generator=torch.Generator().manual_seed(4))
    M_total, f_off = Fr + 4, 3
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.layer_norm(xr, (D,), wr, br, 1e-5) * (toks != PAD).double()[:, None]
    dout = rnd(B, M_total, D, seed=5)
    yr.backward(dout[:, f_off:f_off + Fr].reshape(B * Fr, D).double())
    out = torch.zeros(B, M_total, D, device=dev)
    xhat, rstd = torch.empty(B * Fr, D, device=dev), torch.empty(B * Fr, device=dev)
    ops.layernorm_fwd(x.to(dev), w.to(dev), b.to(dev), out, xhat, rstd, 1e-5, Fr, M_total, f_off, 0.0, 0, None,
                      toks.to(dev))
    assert err(out[:, f_off:f_off + Fr].reshape(B * Fr, D), yr) < 2e-5
    assert float(out[:, :f_off].abs().max()) == 0 and float(out[:, f_off + Fr:].abs().max()) == 0
    dx = torch.empty(B * Fr, D, device=dev)
    dw, db = torch.full((D,), 0.5, device=dev), torch.full((D,), -0.25, device=dev)
    ops.layernorm_bwd(dout.to(dev), xhat, rstd, w.to(dev), dx, dw, db, Fr, M_total, f_off, 0.0, 0, None, toks.to(dev),
                      beta=1.0)
    assert err(dx, xr.grad) < 5e-5
    assert err(dw - 0.5, wr.grad) < 1e-4 and err(db + 0.25, br.grad) < 1e-4
    # inference: no saved statistics
    out2 = torch.zeros(B * Fr, D, device=dev)
    ops.layernorm_fwd(x.to(dev), w.to(dev), b.to(dev), out2, None, None, 1e-5, B * Fr, B * Fr, 0, 0.0, 0, None, None)
    assert err(out2, F.layer_norm(x.double(), (D,), w.double(), b.double(), 1e-5)) < 2e-5


def test_layernorm_dropout_mask_replays_in_backward(dev):
    ops, _ = _ops()
    rows, D, p = 64, 128, 0.3
    x, w, b = rnd(rows, D, seed=1).to(dev), (rnd(D, seed=2) + 1.5).to(dev), rnd(D, seed=3).to(dev)
    rng = ops.RngState(123, dev)
    out, xhat, rstd = torch.empty(rows, D, device=dev), torch.empty(rows, D, device=dev), torch.empty(rows, device=dev)
    ops.layernorm_fwd(x, w, b, out, xhat, rstd, 1e-5, rows, rows, 0, p, 9, rng, None)
    ref = F.layer_norm(x, (D,), w, b, 1e-5)
    keep = out != 0
    frac = float(keep.float().mean())
    assert abs(frac - (1 - p)) < 0.03
    assert err(out[keep], ref[keep] / (1 - p)) < 2e-5
    # backward applies the same mask: d/dx of sum(out) == d/dx of sum(ref * keep/(1-p))
    xr = x.double().cpu().requires_grad_(True)
    (F.layer_norm(xr, (D,), w.double().cpu(), b.double().cpu(), 1e-5) * keep.cpu().double() / (1 - p)).sum().backward()
    dx, dw, db = torch.empty(rows, D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    ops.layernorm_bwd(torch.ones(rows, D, device=dev), xhat, rstd, w, dx, dw, db, rows, rows, 0, p, 9, rng, None)
    assert err(dx, xr.grad) < 5e-5


def test_loss_combine_fwd_bwd(dev):
    """criterion tail: weighted total + meter accumulation in one launch, and its gradient slab"""
    ops, _ = _ops()
    S, n = 8, 3
    slab = rnd(n * S, seed=1).to(dev)
    coef = torch.tensor([0.25, 1.5, -2.0], device=dev)
    total = torch.empty(1, device=dev)
    m_dst = torch.tensor([0, 0, 1, 2, 3], dtype=torch.int32, device=dev)
    m_src = torch.tensor([0, S, 1, 2, 2 * S], dtype=torch.int32, device=dev)
    m_scale = torch.tensor([0.8, 1.0, 1.0, 1.0, 64.0], device=dev)
    meters = torch.tensor([10.0, 20.0, 30.0, 40.0], device=dev)
    ops.loss_combine(slab, n, S, coef, total, m_dst, m_src, m_scale, meters)
    sl = slab.cpu().double()
    assert abs(float(total) - float(0.25 * sl[0] + 1.5 * sl[S] - 2.0 * sl[2 * S])) < 1e-6
    want = [10 + 0.8 * sl[0] + sl[S], 20 + sl[1], 30 + sl[2], 40 + 64 * sl[2 * S]]
    assert err(meters, torch.tensor([float(w) for w in want])) < 1e-5
    ops.loss_combine(slab, n, S, coef, total, None, None, None, None)      # no meter table
    g = torch.tensor([3.0], device=dev)
    gslab = torch.full((n * S,), float("nan"), device=dev)
    ops.loss_combine_bwd(g, coef, n, S, gslab)
    ref = torch.zeros(n * S)
    ref[0], ref[S], ref[2 * S] = 0.75, 4.5, -6.0
    assert torch.equal(gslab.cpu(), ref)


def _bf(x):
    return x.float().to(torch.bfloat16).double()


def _mha_bf16_restatement(q, k, v, do, H, key_pad, causal):
    """the throughput mode's attention written out in float64: every matrix-instruction operand rounded to bf16 (nearest even), the
    soft-max, its backward and every sum in full precision -- attn_mfma.hpp "PR = 1".  q [R, Lq, D], k / v [R, Lk, D] (already
    gathered per sequence); returns o, dq, dk, dv with dk / dv per sequence (the caller sums sequences that share a memory)."""
    R, Lq, D = q.shape
    Lk, dk = k.shape[1], D // H
    sp = lambda x, L: x.view(R, L, H, dk).permute(0, 2, 1, 3)
    qh, kh, vh, doh = sp(q, Lq), sp(k, Lk), sp(v, Lk), sp(do, Lq)
    s = _bf(qh) @ _bf(kh).transpose(-1, -2) / math.sqrt(dk)
    m = torch.zeros(R, 1, Lq, Lk, dtype=torch.bool)
    if key_pad is not None:
        m = m | key_pad.view(R, 1, 1, Lk)
    if causal:
        m = m | torch.triu(torch.ones(Lq, Lk, dtype=torch.bool), 1).view(1, 1, Lq, Lk)
    p = torch.softmax(s.masked_fill(m, -10e6), -1)
    o = _bf(p) @ _bf(vh)
    dp = _bf(doh) @ _bf(vh).transpose(-1, -2)
    ds = p * (dp - (p * dp).sum(-1, keepdim=True)) / math.sqrt(dk)
    dq = _bf(ds) @ _bf(kh)
    dkk = _bf(ds).transpose(-1, -2) @ _bf(qh)
    dv = _bf(p).transpose(-1, -2) @ _bf(doh)
    back = lambda x, L: x.permute(0, 2, 1, 3).reshape(R, L, D)
    return back(o, Lq), back(dq, Lq), back(dkk, Lk), back(dv, Lk)


@pytest.mark.parametrize("kind,Lq,Lk", [("self", 20, 20), ("self_causal", 20, 20), ("cross", 20, 120), ("cross", 32, 70),
                                        ("cross_beams", 20, 120)])
def test_attention_with_bf16_operands_equals_its_rounding_restatement(dev, kind, Lq, Lk, monkeypatch):
    """NACF_ATTN_BF16=1 (the default of the bf16 GEMM mode at dk = 64): v_mfma_f32_16x16x16_bf16 on operands rounded in registers.
    Every kernel of the family (forward, LDS-staged forward, backward, key-block backward) against the float64 restatement that
    rounds the same operands; the bar is what one bf16 rounding of a P / dS element that sits on a rounding boundary can move
    (the kernel's fp32 soft-max and the restatement's differ in the last bits), far below the distance to the unrounded result."""
    monkeypatch.setenv("NACF_ATTN_BF16", "1")
    ops, _ = _ops()
    H, dk = 8, 64
    D = H * dk
    if kind.startswith("self"):
        R, Bv, kv_div, kv_mod = 6, 6, 1, 6
        causal = kind == "self_causal"
        tok = torch.randint(1, 9, (R, Lq), generator=torch.Generator().manual_seed(2))
        tok[0, Lq - 3:] = PAD
        tok[1, 2] = PAD
        vid = torch.arange(R)
    else:
        k_ = 6 if kind == "cross_beams" else 2            # 6 sequences per memory: the LDS-staged forward
        Bv = 5
        R, kv_div, kv_mod, causal, tok = Bv * k_, k_, Bv, False, None
        vid = torch.arange(R) // k_
    q, kv, do = rnd(R * Lq, D, seed=1), rnd(Bv * Lk, 2 * D, seed=2), rnd(R * Lq, D, seed=3)
    kk, vv = kv[:, :D].reshape(Bv, Lk, D)[vid], kv[:, D:].reshape(Bv, Lk, D)[vid]
    o_ref, dq_ref, dk_seq, dv_seq = _mha_bf16_restatement(q.view(R, Lq, D).double(), kk.double(), vv.double(),
                                                           do.view(R, Lq, D).double(), H, None if tok is None else tok.eq(PAD), causal)
    dk_ref = torch.zeros(Bv, Lk, D, dtype=torch.float64).index_add_(0, vid, dk_seq)
    dv_ref = torch.zeros(Bv, Lk, D, dtype=torch.float64).index_add_(0, vid, dv_seq)
    o_exact, _ = _mha_ref(q.view(R, Lq, D).double(), kk.double(), vv.double(), H, None if tok is None else tok.eq(PAD), causal)
    qx, kvx, dox = q.to(dev), kv.to(dev), do.to(dev)
    tokx = None if tok is None else tok.to(dev)
    out = torch.full_like(qx, float("nan"))
    ops.attention_fwd(qx, kvx[:, :D], kvx[:, D:], out, tokx, int(causal), None, R, H, Lq, Lk, dk, kv_div, kv_mod)
    dq, dkv = torch.full_like(qx, float("nan")), torch.full_like(kvx, float("nan"))
    ops.attention_bwd(qx, kvx[:, :D], kvx[:, D:], dox, dq, dkv[:, :D], dkv[:, D:], tokx, int(causal), R, Bv, H, Lq, Lk, dk,
                      kv_div, kv_mod)
    e_o = err(out.view(R, Lq, D), o_ref)
    assert e_o < 1e-3, e_o
    assert err(out.view(R, Lq, D), o_exact) > 4 * e_o            # it IS the bf16 form
    assert err(dq.view(R, Lq, D), dq_ref) < 2e-3
    assert err(dkv[:, :D].reshape(Bv, Lk, D), dk_ref) < 4e-3 and err(dkv[:, D:].reshape(Bv, Lk, D), dv_ref) < 4e-3
    monkeypatch.setenv("NACF_ATTN_BF16", "0")                    # and the switch is per call: fp32 operands again
    out32 = torch.empty_like(qx)
    ops.attention_fwd(qx, kvx[:, :D], kvx[:, D:], out32, tokx, int(causal), None, R, H, Lq, Lk, dk, kv_div, kv_mod)
    assert err(out32.view(R, Lq, D), o_exact) < 2e-5


@pytest.mark.parametrize("Lk,Lq", [(120, 20), (120, 32), (70, 7), (33, 1)])
def test_key_block_backward_batched_loads_change_no_bit(dev, Lk, Lq, monkeypatch):
    """the key-block kernel with its operand loads issued as branch-free batches (rows past the end read the last live row and are
    replaced by zero) against its first form with guarded loads (NACF_ATTN_KB=2): the same operations on the same values"""
    ops, _ = _ops()
    Bv, k_, H, dk = 6, 2, 8, 64
    D, R = H * dk, Bv * k_
    q, kv, do = rnd(R * Lq, D, seed=1).to(dev), rnd(Bv * Lk, 2 * D, seed=2).to(dev), rnd(R * Lq, D, seed=3).to(dev)
    outs = {}
    for kb in ("2", "1"):
        monkeypatch.setenv("NACF_ATTN_KB", kb)
        dq, dkv = torch.full_like(q, float("nan")), torch.full_like(kv, float("nan"))
        ops.attention_bwd(q, kv[:, :D], kv[:, D:], do, dq, dkv[:, :D], dkv[:, D:], None, 0, R, Bv, H, Lq, Lk, dk, k_, Bv)
        outs[kb] = (dq, dkv)
    assert torch.equal(outs["1"][0], outs["2"][0]) and torch.equal(outs["1"][1], outs["2"][1])
    assert not bool(torch.isnan(outs["1"][0]).any()) and not bool(torch.isnan(outs["1"][1]).any())


@pytest.mark.parametrize("kb", ["1", "0", "2"])
@pytest.mark.parametrize("mode,Lk", [("mod", 120), ("div", 120), ("mod", 70), ("mod", 33)])
def test_cross_attention_backward_model_width(dev, mode, Lk, kb, monkeypatch):
    """dk = 64, memory of up to 128 keys: the key-block kernel (four waves split the keys of a (video, head); softmax
    statistics, delta and dQ combined through LDS) and the one-wave-per-item kernel against float64 autograd"""
    monkeypatch.setenv("NACF_ATTN_KB", kb)
    ops, _ = _ops()
    Bv, k_, Lq, H, dk = 5, 2, 20, 8, 64
    D, R = H * dk, Bv * k_
    q = rnd(R * Lq, D, seed=1)
    kv = rnd(Bv * Lk, 2 * D, seed=2)
    vid = (torch.arange(R) % Bv) if mode == "mod" else (torch.arange(R) // k_)
    kv_div, kv_mod = (1, Bv) if mode == "mod" else (k_, Bv)
    qd, kvd = q.double().requires_grad_(True), kv.double().requires_grad_(True)
    o_ref, _ = _mha_ref(qd.view(R, Lq, D), kvd[:, :D].reshape(Bv, Lk, D)[vid], kvd[:, D:].reshape(Bv, Lk, D)[vid], H, None, False)
    do = rnd(R * Lq, D, seed=3)
    o_ref.backward(do.view(R, Lq, D).double())
    qx, kvx = q.to(dev), kv.to(dev)
    dq, dkv = torch.full_like(qx, float("nan")), torch.full_like(kvx, float("nan"))
    ops.attention_bwd(qx, kvx[:, :D], kvx[:, D:], do.to(dev), dq, dkv[:, :D], dkv[:, D:], None, 0, R, Bv, H, Lq, Lk, dk,
                      kv_div, kv_mod)
    assert err(dq, qd.grad) < 5e-5 and err(dkv, kvd.grad) < 1e-4
    again_q, again_kv = torch.empty_like(qx), torch.empty_like(kvx)
    ops.attention_bwd(qx, kvx[:, :D], kvx[:, D:], do.to(dev), again_q, again_kv[:, :D], again_kv[:, D:], None, 0, R, Bv, H,
                      Lq, Lk, dk, kv_div, kv_mod)
    assert torch.equal(dq, again_q) and torch.equal(dkv, again_kv)        # fixed combination order: run to run identical


@pytest.mark.parametrize("rows,V,K", [(70, 101, 64), (300, 10547, 512)])
def test_vocab_projection_with_fused_softmax_statistics(dev, rows, V, K):
    """nacf_vocab_lse_fwd: raw logits stored by the GEMM epilogue + lse / argmax / log p(label) from its per-tile
    partials, live rows only; nacf_xent_bwd_lse turns the logits into the cross-entropy gradient in place"""
    ops, _ = _ops()
    h, w, b = rnd(rows, K, seed=1), rnd(V, K, seed=2, scale=0.5), rnd(V, seed=3)
    labels = torch.randint(0, min(V, 50), (rows,), generator=torch.Generator().manual_seed(4))
    labels[::5] = PAD
    z_ref = h.double() @ w.double().t() + b.double()
    lse_ref = torch.logsumexp(z_ref, dim=1)
    live = ops.rowset_build(tokens=labels.to(dev))
    buf = torch.full((rows, ops.vocab_ld(V)), float("nan"), device=dev)
    logits = buf[:, :V]
    lse, llp = torch.zeros(rows, device=dev), torch.zeros(rows, device=dev)
    am = torch.zeros(rows, dtype=torch.int64, device=dev)
    ops.vocab_lse_fwd(h.to(dev), w.to(dev), b.to(dev), logits, labels.to(dev), lse, am, llp, live)
    keep = labels != PAD
    assert err(logits[keep.to(dev)], z_ref[keep]) < 2e-4
    assert err(lse[keep.to(dev)], lse_ref[keep]) < 1e-4
    assert torch.equal(am.cpu()[keep], z_ref.argmax(1)[keep])
    assert err(llp[keep.to(dev)], (z_ref.gather(1, labels.view(-1, 1)).squeeze(1) - lse_ref)[keep]) < 2e-4
    assert bool(torch.isnan(logits[(~keep).to(dev)]).all())              # rows without a label are not touched
    g = torch.tensor([0.37], device=dev)
    ops.xent_bwd_lse(logits, lse, logits, V, labels.to(dev), g, 1.0, skip_pad_rows=True)
    want = (torch.softmax(z_ref, 1) - torch.nn.functional.one_hot(labels, V).double()) * 0.37
    assert err(logits[keep.to(dev)], want[keep]) < 5e-6


# ------------------------------------------------------------------ the exact GEMM shapes of the bench step (B=128)
def _close64(got, ref, K, what):
    """fp32 accumulation over K terms of O(1) data vs fp64"""
    tol = 2e-6 * math.sqrt(K) * max(1.0, float(ref.abs().max()))
    e = err(got, ref)
    assert e < tol, (what, e, tol)


def test_bench_shape_vocab_gemms_vs_fp64(dev):
    """vocabulary projection at the bench shape (models/__init__.py:83; 2B*L = 5120 slots, V = 10547, D = 512, about
    45 % of them labelled): forward logits + soft-max statistics, dX (reduce over V, split-K) and dW (reduce over the
    live rows, split-K) against fp64 on the host, with the production tile / split heuristics"""
    ops, _ = _ops()
    M, V, K = 5120, 10547, 512
    g = torch.Generator().manual_seed(0)
    h, w = rnd(M, K, seed=1, scale=0.5), rnd(V, K, seed=2, scale=0.1)
    labels = torch.randint(6, V, (M,), generator=g)
    labels[torch.rand(M, generator=g) < 0.55] = PAD
    live_idx = labels.ne(PAD).nonzero().squeeze(1)
    hd, wd, ld = h.to(dev), w.to(dev), labels.to(dev)
    live = ops.rowset_build(tokens=ld)
    buf = torch.empty(M, ops.vocab_ld(V), device=dev)
    logits = buf[:, :V]
    lse, llp = torch.empty(M, device=dev), torch.empty(M, device=dev)
    am = torch.empty(M, dtype=torch.int64, device=dev)
    ops.vocab_lse_fwd(hd, wd, None, logits, ld, lse, am, llp, live)
    ref = h[live_idx].double() @ w.double().t()
    _close64(logits[live_idx.to(dev)], ref, K, "vocab fwd")
    assert err(lse[live_idx.to(dev)], torch.logsumexp(ref, 1)) < 2e-5
    # backward operands: a dense-looking dlogits on the live rows, garbage (NaN) on the dead ones -- never read
    dz = torch.full((M, ops.vocab_ld(V)), float("nan"))
    dzl = rnd(live_idx.numel(), V, seed=3, scale=0.01)
    dz[live_idx, :V] = dzl
    dzd = dz.to(dev)[:, :V]
    dx = torch.empty(M, K, device=dev)
    ops.linear_bwd_data(dzd, wd, dx, rows=live, zero_dead=True)
    _close64(dx[live_idx.to(dev)], dzl.double() @ w.double(), V, "vocab dX")
    dead = labels.eq(PAD).to(dev)
    assert float(dx[dead].abs().max()) == 0.0
    dw = torch.zeros(V, K, device=dev)
    ops.linear_bwd_weight(dzd, hd, dw, None, beta=0.0, rows=live)
    _close64(dw, dzl.double().t() @ h[live_idx].double(), live_idx.numel(), "vocab dW")


@pytest.mark.parametrize("M,N,K,what", [(7680, 512, 2048, "encoder Linear (models/Encoder.py:62)"),
                                        (15360, 1024, 512, "cross K|V of the memory (models/bert.py:146-148)"),
                                        (7680, 1024, 512, "HighWay w1|w2 (models/Encoder.py:19-25)"),
                                        (5120, 2048, 512, "FFN up (models/bert.py:227-230)"),
                                        (5120, 512, 2048, "FFN down (models/bert.py:240-247)")])
def test_bench_shape_linear_fwd_dx_dw_vs_fp64(dev, M, N, K, what):
    """every dense Linear of the B=128 step at its real size: forward, dX and dW (+ bias gradient) vs fp64"""
    ops, _ = _ops()
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3)
    dz = rnd(M, N, seed=4, scale=0.05)
    xd, wd, dzd = x.to(dev), w.to(dev), dz.to(dev)
    y = torch.empty(M, N, device=dev)
    ops.linear_fwd(xd, wd, y, ops.Epi(bias=b.to(dev)))
    _close64(y, x.double() @ w.double().t() + b.double(), K, what + " fwd")
    dx = torch.empty(M, K, device=dev)
    ops.linear_bwd_data(dzd, wd, dx)
    _close64(dx, dz.double() @ w.double(), N, what + " dX")
    dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    ops.linear_bwd_weight(dzd, xd, dw, db, beta=0.0)
    _close64(dw, dz.double().t() @ x.double(), M, what + " dW")
    _close64(db, dz.double().sum(0), M, what + " db")


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("B,Fr,D", [(8, 6, 64), (16, 60, 512), (8, 7, 100)])
def test_sync_bn_kernels_equal_the_global_batch(dev, B, Fr, D, world):
    """data-parallel BatchNorm (nacf_bn_sync_stat / _concat_fwd_sync / _sync_bwd_stat / _concat_bwd_sync): the batch is
    cut into `world` shards that are processed as `world` ranks would, the exchanged vectors are summed by hand (the
    all-reduce), and every shard's output, dx and the summed dw / db must equal the single-process kernels on the
    whole batch -- and fp64 BatchNorm (models/joint_representation.py:43-45)"""
    ops, _ = _ops()
    M_total, f_off = Fr + 5, 3
    x = rnd(B, Fr, D, seed=1).to(dev) * 2 + 0.5
    w, b = (rnd(D, seed=2) + 1.5).to(dev), rnd(D, seed=3).to(dev)
    dout = rnd(B, M_total, D, seed=4).to(dev)
    # single process, whole batch
    out0 = torch.zeros(B, M_total, D, device=dev)
    rm0, rv0 = torch.zeros(D, device=dev), torch.ones(D, device=dev)
    nbt0 = torch.zeros((), dtype=torch.int64, device=dev)
    sm0, si0 = torch.empty(D, device=dev), torch.empty(D, device=dev)
    ops.bn_concat_fwd(x, out0, f_off, w, b, rm0, rv0, nbt0, sm0, si0, True)
    dx0, dw0, db0 = torch.empty_like(x), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    ops.bn_concat_bwd(dout, x, dx0, f_off, w, sm0, si0, dw0, db0, beta=0.0)
    # `world` ranks
    per = B // world
    xs = [x[r * per:(r + 1) * per].contiguous() for r in range(world)]
    douts = [dout[r * per:(r + 1) * per].contiguous() for r in range(world)]
    n_total = B * Fr
    S_loc = [torch.empty(D, device=dev) for _ in range(world)]
    for r in range(world):
        ops.bn_sync_stat(xs[r], None, n_total, S_loc[r])
    S = torch.stack(S_loc).sum(0)                                   # all-reduce
    Q_loc = [torch.empty(D, device=dev) for _ in range(world)]
    for r in range(world):
        ops.bn_sync_stat(xs[r], S, n_total, Q_loc[r])
    Q = torch.stack(Q_loc).sum(0)                                   # all-reduce
    # ONE exchange instead (the default protocol): per rank (sum | squared deviations about its OWN mean), all-gathered,
    # merged by nacf_bn_sync_merge -- the same S, and Q to fp32 round-off of the two-pass Q and of fp64
    n_loc = per * Fr
    gathered = torch.empty(world, 2, 1, D, device=dev)
    for r in range(world):
        ops.bn_sync_stat(xs[r], None, n_loc, gathered[r, 0, 0])
        ops.bn_sync_stat(xs[r], gathered[r, 0, 0], n_loc, gathered[r, 1, 0])
    merged = torch.empty(2, 1, D, device=dev)
    ops.bn_sync_merge(gathered, [n_loc], merged)
    x64 = x.double().reshape(B * Fr, D)
    Q64 = ((x64 - x64.mean(0)) ** 2).sum(0)
    assert err(merged[0, 0], S) <= 2e-6 * float(S.abs().max())
    assert err(merged[1, 0], Q64) <= 3e-6 * float(Q64.max()) and err(Q, Q64) <= 3e-6 * float(Q64.max())
    # the row counts travel behind every rank's statistics: equal counts merge to the same bits and leave the flag alone; a rank with
    # another count (a ragged global batch) turns the merged statistics into NaN and raises the sticky flag
    C2 = 2 * D
    ext = torch.zeros(world, C2 + 1, device=dev)
    ext[:, :C2] = gathered.view(world, C2)
    ext[:, C2] = float(n_loc)
    merged2, flag = torch.empty(2, 1, D, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    ops.bn_sync_merge(ext, [n_loc], merged2, flag)
    assert torch.equal(merged2, merged) and int(flag.item()) == 0
    ext[1, C2] = float(n_loc - 3)
    ops.bn_sync_merge(ext, [n_loc], merged2, flag)
    assert bool(torch.isnan(merged2).all()) and int(flag.item()) == 1
    S, Q = merged[0, 0].clone(), merged[1, 0].clone()               # what the ranks apply from here on
    outs, saves, stats = [], [], []
    for r in range(world):
        o = torch.zeros(per, M_total, D, device=dev)
        rm, rv = torch.zeros(D, device=dev), torch.ones(D, device=dev)
        nbt = torch.zeros((), dtype=torch.int64, device=dev)
        sm, si = torch.empty(D, device=dev), torch.empty(D, device=dev)
        ops.bn_concat_fwd_sync(xs[r], o, f_off, w, b, rm, rv, nbt, sm, si, S, Q, n_total)
        outs.append(o); saves.append((sm, si)); stats.append((rm, rv, nbt))
    assert err(torch.cat(outs), out0) < 2e-5
    for rm, rv, nbt in stats:                                       # every rank ends with the GLOBAL running statistics
        assert err(rm, rm0) < 1e-6 and err(rv, rv0) < 1e-5 and int(nbt) == 1
    assert float(out0[:, :f_off].abs().max()) == 0 and float(torch.cat(outs)[:, :f_off].abs().max()) == 0
    sums_loc, dws, dbs = [], [], []
    for r in range(world):
        s2 = torch.empty(2, D, device=dev)
        dw, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
        ops.bn_sync_bwd_stat(douts[r], xs[r], f_off, saves[r][0], saves[r][1], s2, dw, db, beta=0.0)
        sums_loc.append(s2); dws.append(dw); dbs.append(db)
    sums = torch.stack(sums_loc).sum(0)                             # all-reduce
    dxs = []
    for r in range(world):
        dx = torch.empty_like(xs[r])
        ops.bn_concat_bwd_sync(douts[r], xs[r], dx, f_off, w, saves[r][0], saves[r][1], sums, n_total)
        dxs.append(dx)
    scale = float(dx0.abs().max())
    assert err(torch.cat(dxs), dx0) < 2e-5 * max(1.0, scale)
    assert err(torch.stack(dws).sum(0), dw0) < 1e-4 * max(1.0, float(dw0.abs().max()))      # gradient all-reduce adds the ranks
    assert err(torch.stack(dbs).sum(0), db0) < 1e-4 * max(1.0, float(db0.abs().max()))
    # and against fp64 BatchNorm on the whole batch
    xd = x.double().cpu().reshape(B * Fr, D).requires_grad_(True)
    y = F.batch_norm(xd, None, None, w.double().cpu(), b.double().cpu(), True, 0.1, 1e-5)
    assert err(torch.cat(outs)[:, f_off:f_off + Fr].reshape(B * Fr, D), y) < 5e-5
    y.backward(dout[:, f_off:f_off + Fr].double().cpu().reshape(B * Fr, D))
    assert err(torch.cat(dxs).reshape(B * Fr, D), xd.grad) < 5e-5 * max(1.0, float(xd.grad.abs().max()))


def test_bn_concat_of_all_modalities_in_one_launch(dev):
    """nacf_bn_concat_fwd_multi / _bwd_multi: the modalities of a joint representation side by side in the same launches --
    bit-identical to one nacf_bn_concat_fwd / _bwd call per modality (different frame counts, running statistics,
    parameter gradients accumulated with beta = 1), training and eval"""
    ops, _ = _ops()
    B, D, Fs = 24, 192, [7, 12, 5]
    M_total = sum(Fs) + 2
    f_offs = [1, 1 + Fs[0], 1 + Fs[0] + Fs[1]]
    xs = [(rnd(B, f, D, seed=10 + i) * (1 + i) + 0.3 * i).to(dev) for i, f in enumerate(Fs)]
    ws = [(rnd(D, seed=20 + i) + 1.5).to(dev) for i in range(3)]
    bs = [rnd(D, seed=30 + i).to(dev) for i in range(3)]
    dout = rnd(B, M_total, D, seed=4).to(dev)

    def fresh():
        return ([torch.full((D,), 0.1 * i, device=dev) for i in range(3)], [torch.full((D,), 1.0 + i, device=dev) for i in range(3)],
                [torch.zeros((), dtype=torch.int64, device=dev) for _ in range(3)],
                [torch.empty(D, device=dev) for _ in range(3)], [torch.empty(D, device=dev) for _ in range(3)])
    for training in (False, True):           # (training last: its saved statistics feed the backward below)
        rm1, rv1, nbt1, sm1, si1 = fresh()
        out1 = torch.zeros(B, M_total, D, device=dev)
        for i in range(3):
            ops.bn_concat_fwd(xs[i], out1, f_offs[i], ws[i], bs[i], rm1[i], rv1[i], nbt1[i], sm1[i], si1[i], training)
        rm2, rv2, nbt2, sm2, si2 = fresh()
        out2 = torch.zeros(B, M_total, D, device=dev)
        ops.bn_concat_fwd_multi(xs, out2, f_offs, ws, bs, rm2, rv2, nbt2, sm2 if training else [None] * 3,
                                si2 if training else [None] * 3, training)
        assert torch.equal(out1, out2)
        for i in range(3):
            assert torch.equal(rm1[i], rm2[i]) and torch.equal(rv1[i], rv2[i]) and int(nbt1[i]) == int(nbt2[i]) == int(training)
            if training:
                assert torch.equal(sm1[i], sm2[i]) and torch.equal(si1[i], si2[i])
    dws1 = [rnd(D, seed=40 + i).to(dev) for i in range(3)]
    dbs1 = [rnd(D, seed=50 + i).to(dev) for i in range(3)]
    dws2, dbs2 = [t.clone() for t in dws1], [t.clone() for t in dbs1]
    dx1, dx2 = [torch.empty_like(x) for x in xs], [torch.empty_like(x) for x in xs]
    for i in range(3):
        ops.bn_concat_bwd(dout, xs[i], dx1[i], f_offs[i], ws[i], sm1[i], si1[i], dws1[i], dbs1[i], beta=1.0)
    ops.bn_concat_bwd_multi(dout, xs, dx2, f_offs, ws, sm1, si1, dws2, dbs2, beta=1.0)
    for i in range(3):
        assert torch.equal(dx1[i], dx2[i]) and torch.equal(dws1[i], dws2[i]) and torch.equal(dbs1[i], dbs2[i])


def test_sync_bn_of_all_modalities_in_one_launch(dev):
    """the data-parallel BatchNorm entry points for every modality at once (nacf_bn_sync_local_multi, _sync_bwd_local_multi,
    the stats_global / sums_global forms of nacf_bn_concat_{fwd,bwd}_multi) against the per-modality ones: bit-identical"""
    ops, _ = _ops()
    B, D, Fs, world = 16, 128, [9, 4], 4
    M_total = sum(Fs)
    f_offs = [0, Fs[0]]
    xs = [(rnd(B, f, D, seed=10 + i) * (1 + i) + 0.3 * i).to(dev) for i, f in enumerate(Fs)]
    ws = [(rnd(D, seed=20 + i) + 1.5).to(dev) for i in range(2)]
    bs = [rnd(D, seed=30 + i).to(dev) for i in range(2)]
    dout = rnd(B, M_total, D, seed=4).to(dev)
    n_loc = [B * f for f in Fs]
    n_tot = [n * world for n in n_loc]
    loc1 = torch.empty(2, 2, D, device=dev)
    for i in range(2):
        ops.bn_sync_stat(xs[i], None, n_loc[i], loc1[0, i])
        ops.bn_sync_stat(xs[i], loc1[0, i], n_loc[i], loc1[1, i])
    loc2 = torch.empty(2, 2, D, device=dev)
    ops.bn_sync_local_multi(xs, loc2)
    assert torch.equal(loc1, loc2)
    # pretend `world` ranks hold shifted copies of these rows: any gathered vector will do for the comparison
    gathered = torch.stack([loc1 * (1 + 0.01 * r) for r in range(world)]).contiguous()
    stats = torch.empty(2, 2, D, device=dev)
    ops.bn_sync_merge(gathered, n_loc, stats)

    def fresh():
        return ([torch.zeros(D, device=dev) for _ in range(2)], [torch.ones(D, device=dev) for _ in range(2)],
                [torch.zeros((), dtype=torch.int64, device=dev) for _ in range(2)],
                [torch.empty(D, device=dev) for _ in range(2)], [torch.empty(D, device=dev) for _ in range(2)])
    rm1, rv1, nbt1, sm1, si1 = fresh()
    out1 = torch.zeros(B, M_total, D, device=dev)
    for i in range(2):
        ops.bn_concat_fwd_sync(xs[i], out1, f_offs[i], ws[i], bs[i], rm1[i], rv1[i], nbt1[i], sm1[i], si1[i], stats[0, i], stats[1, i], n_tot[i])
    rm2, rv2, nbt2, sm2, si2 = fresh()
    out2 = torch.zeros(B, M_total, D, device=dev)
    ops.bn_concat_fwd_multi(xs, out2, f_offs, ws, bs, rm2, rv2, nbt2, sm2, si2, True, stats_global=stats, n_total=n_tot)
    assert torch.equal(out1, out2)
    for i in range(2):
        assert torch.equal(rm1[i], rm2[i]) and torch.equal(rv1[i], rv2[i]) and torch.equal(sm1[i], sm2[i]) and torch.equal(si1[i], si2[i])
    dws1, dbs1 = [rnd(D, seed=40 + i).to(dev) for i in range(2)], [rnd(D, seed=50 + i).to(dev) for i in range(2)]
    dws2, dbs2 = [t.clone() for t in dws1], [t.clone() for t in dbs1]
    sums1, sums2 = torch.empty(2, 2, D, device=dev), torch.empty(2, 2, D, device=dev)
    for i in range(2):
        ops.bn_sync_bwd_stat(dout, xs[i], f_offs[i], sm1[i], si1[i], sums1[i], dws1[i], dbs1[i], beta=1.0)
    ops.bn_sync_bwd_local_multi(dout, xs, f_offs, sm1, si1, sums2, dws2, dbs2, beta=1.0)
    assert torch.equal(sums1, sums2)
    for i in range(2):
        assert torch.equal(dws1[i], dws2[i]) and torch.equal(dbs1[i], dbs2[i])
    sums_g = (sums1 * world).contiguous()
    dx1, dx2 = [torch.empty_like(x) for x in xs], [torch.empty_like(x) for x in xs]
    for i in range(2):
        ops.bn_concat_bwd_sync(dout, xs[i], dx1[i], f_offs[i], ws[i], sm1[i], si1[i], sums_g[i], n_tot[i])
    keep_w, keep_b = [t.clone() for t in dws2], [t.clone() for t in dbs2]
    ops.bn_concat_bwd_multi(dout, xs, dx2, f_offs, ws, sm1, si1, dws2, dbs2, beta=1.0, sums_global=sums_g, n_total=n_tot)
    for i in range(2):
        assert torch.equal(dx1[i], dx2[i]) and torch.equal(dws2[i], keep_w[i]) and torch.equal(dbs2[i], keep_b[i])


def test_adam_walk_can_leave_the_gradient_zeroed(dev):
    """nacf_adam_step_part, bump bit 2: the same update, and the gradient slice is zero afterwards (runtime/engine.py skips
    the next step's fill); without the bit the gradient is left as optimizer.step() of the reference leaves it"""
    ops, _ = _ops()
    n = 100_003
    p0, g0 = rnd(n, seed=1).to(dev), (rnd(n, seed=2) * 8).to(dev)
    res = []
    for z in (False, True):
        p, g = p0.clone(), g0.clone()
        m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        lr, step = torch.full((1,), 1e-3, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
        ops.adam_step(p[:70000], g[:70000], m[:70000], v[:70000], lr, step, 0.9, 0.999, 1e-8, 5e-4, 5.0, 0.5, bump=True, zero_grad=z)
        res.append((p, g, m, v, int(step)))
    (pa, ga, ma, va, sa), (pb, gb, mb, vb, sb) = res
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb) and sa == sb == 1
    assert torch.equal(ga, g0) and float(gb[:70000].abs().max()) == 0.0 and torch.equal(gb[70000:], g0[70000:])


def test_deferred_dw_combines_are_bit_identical(dev):
    """ops.dw_group(): the split-K combines of many weight-gradient GEMMs run in one launch at the end -- same sums bit
    for bit as the per-call combine; a second gradient for the same dW is serialised; more than 32 queued combines
    flush themselves; the slabs of every queued combine survive until the flush (own workspace slices)."""
    from nacf_amd.runtime import lib as L, ops
    g = torch.Generator().manual_seed(3)
    r = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).to(dev)
    shapes = [(2980, 512, 512), (2980, 2048, 512), (2980, 512, 2048), (7680, 512, 2048), (5120, 1536, 512), (777, 96, 200),
              (3000, 20, 512)]
    probs = [(r(M, N), r(M, K), r(N, K), r(N)) for M, N, K in shapes]
    ref = []
    for dz, x, w0, b0 in probs:
        dw, db = w0.clone(), b0.clone()
        ops.linear_bwd_weight(dz, x, dw, db, beta=1.0)
        ops.linear_bwd_weight(dz, x, dw, db, beta=1.0)          # the layer applied twice: accumulates
        ref.append((dw, db))
    torch.cuda.synchronize()
    out = [(w0.clone(), b0.clone()) for _, _, w0, b0 in probs]
    with ops.dw_group(defer_gemm=False):
        for (dz, x, _, _), (dw, db) in zip(probs, out):
            ops.linear_bwd_weight(dz, x, dw, db, beta=1.0)
        assert L.load().nacf_dw_group_pending() > 0
        for (dz, x, _, _), (dw, db) in zip(probs, out):          # same targets again: the group restarts by itself
            ops.linear_bwd_weight(dz, x, dw, db, beta=1.0)
    assert L.load().nacf_dw_group_pending() == -1
    torch.cuda.synchronize()
    for (a, b), (c, d), shp in zip(ref, out, shapes):
        assert torch.equal(a, c) and torch.equal(b, d), shp
    # more combines than the queue holds
    many = [(r(1500, 64), r(1500, 64)) for _ in range(40)]
    dws_ref = [torch.zeros(64, 64, device=dev) for _ in many]
    dws = [torch.zeros(64, 64, device=dev) for _ in many]
    for (dz, x), dw in zip(many, dws_ref):
        ops.linear_bwd_weight(dz, x, dw, None, beta=0.0)
    with ops.dw_group(defer_gemm=False):
        for (dz, x), dw in zip(many, dws):
            ops.linear_bwd_weight(dz, x, dw, None, beta=0.0)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(dws_ref, dws))


def test_grouped_dw_gemms(dev):
    """ops.dw_group() with the GEMMs deferred too: all weight-gradient GEMMs of a backward pass in one grid per 16 problems
    (device-side problem table, splits chosen for the group).  Against float64: as accurate as the one-at-a-time launches;
    live-row lists, bias gradients, accumulation (beta = 1), problems the grouped kernel does not take (N < 128: launched
    at once), more than 16 problems, a dW queued twice (the group restarts), and run-to-run bit-identity."""
    import ctypes
    from nacf_amd.runtime import lib as L, ops
    g = torch.Generator().manual_seed(11)
    r = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).to(dev)
    shapes = [(5120, 512, 512), (5120, 2048, 512), (5120, 512, 2048), (7680, 512, 2048), (5120, 1536, 512), (15360, 1024, 512),
              (5120, 10547, 512), (3000, 20, 512), (128, 512, 512), (2000, 130, 260)]
    probs = []
    for i, (M, N, K) in enumerate(shapes):
        dz, x = r(M, ops.vocab_ld(N))[:, :N], r(M, K)
        rows = None
        if i in (0, 1, 6):                                         # decoder-style live-row lists (~58 % live)
            tok = (torch.rand(M, generator=g) < 0.58).to(dev).long()
            rows = ops.rowset_build(tokens=tok)                   # tokens != <pad> (0) are live
        probs.append((dz, x, r(N, K), r(N), rows))
    lib = L.load()

    def run(grouped):
        out = [(w0.clone(), b0.clone()) for _, _, w0, b0, _ in probs]
        if grouped:
            with ops.dw_group():
                for (dz, x, _, _, rows), (dw, db) in zip(probs, out):
                    ops.linear_bwd_weight(dz, x, dw, db, beta=1.0, rows=rows)
                pending = lib.nacf_dw_group_pending()
            nl, nw = ctypes.c_int(0), ctypes.c_int(0)
            lib.nacf_dw_group_stats(ctypes.byref(nl), ctypes.byref(nw))
            seen.append((pending, nl.value, nw.value))
        else:
            for (dz, x, _, _, rows), (dw, db) in zip(probs, out):
                ops.linear_bwd_weight(dz, x, dw, db, beta=1.0, rows=rows)
        torch.cuda.synchronize()
        return out

    seen = []
    one = run(False)
    run(True)                     # sizes the group's slab buffer (a growth flushes what is queued: fewer per launch)
    grp, grp2 = run(True), run(True)
    if ops.gemm_mode() != 0:       # (the fp32-MFMA family launches one GEMM per call; only its combines are grouped)
        assert seen[-1][0] >= 8 and seen[-1][1] == 1 and seen[-1][2] > 0, seen      # 8 of the 10 problems in ONE grid
    for (dz, x, w0, b0, rows), (a, ab), (c, cb), (d, db_) in zip(probs, one, grp, grp2):
        assert torch.equal(c, d) and torch.equal(cb, db_)          # deterministic
        dz64, x64 = dz.double(), x.double()
        if rows is not None:
            live = rows.rows[:int(rows.count)].long()
            dz64, x64 = dz64[live], x64[live]
        ref = w0.double() + dz64.t() @ x64
        refb = b0.double() + dz64.sum(0)
        scale = float(ref.abs().max())
        e_one, e_grp = float((a.double() - ref).abs().max()) / scale, float((c.double() - ref).abs().max()) / scale
        # (a problem that keeps ONE split -- the vocabulary projection on the 256 x 256 body -- sums all its live rows in a single fp32
        #  accumulation chain: ~3000 products, 2.0e-6 of the largest entry measured; the split launches sum shorter chains)
        assert e_grp <= max(3e-6, 1.5 * e_one), (tuple(dz.shape), e_one, e_grp)
        assert float((cb.double() - refb).abs().max()) <= 1e-5 * float(refb.abs().max()) + 1e-4
    # two chunks (more than 16 problems) and a repeated target
    many = [(r(2048, 128), r(2048, 256)) for _ in range(20)]
    ref = [dz.double().t() @ x.double() for dz, x in many]
    dws = [torch.zeros(128, 256, device=dev) for _ in many]
    with ops.dw_group():
        for (dz, x), dw in zip(many, dws):
            ops.linear_bwd_weight(dz, x, dw, None, beta=0.0)
        ops.linear_bwd_weight(many[0][0], many[0][1], dws[0], None, beta=1.0)      # the same dW again: twice the product
    torch.cuda.synchronize()
    ref[0] = 2 * ref[0]
    for a, b in zip(ref, dws):
        assert float((a - b.double()).abs().max()) <= 2e-6 * float(a.abs().max())


@pytest.mark.parametrize("mode", ["bf16x3", "f32"])
def test_small_weight_gradient_kernel(dev, monkeypatch, mode):
    """the length head's second Linear (models/Predictor.py:15-20): dW [20, 512] over 128 rows takes the one-thread-per-element
    kernel (fp32 fmaf chain in row order): fp64 reference, bias gradient, beta = 1, a live-row list, run-to-run identical"""
    ops, L = _ops()
    monkeypatch.setenv("NACF_GEMM_MODE", mode)
    M, N, K = 128, 20, 512
    dz, x = rnd(M, N, seed=1).to(dev), rnd(M, K, seed=2).to(dev)
    w0, b0 = rnd(N, K, seed=3).to(dev), rnd(N, seed=4).to(dev)
    for with_rows in (False, True):
        rows = live = None
        if with_rows:
            tok = (torch.rand(M, generator=torch.Generator().manual_seed(5)) < 0.6).long().to(dev)
            rows, live = ops.rowset_build(tokens=tok), tok.ne(0)
        outs = []
        for _ in range(2):
            dw, db = w0.clone(), b0.clone()
            ops.linear_bwd_weight(dz, x, dw, db, beta=1.0, rows=rows)
            assert L.load().nacf_gemm_last_kernel().decode() == "dw_small_kernel"
            outs.append((dw, db))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        dz64, x64 = dz.double(), x.double()
        if live is not None:
            dz64, x64 = dz64[live], x64[live]
        assert err(outs[0][0], w0.double() + dz64.t() @ x64) < 2e-5
        assert err(outs[0][1], b0.double() + dz64.sum(0)) < 2e-5
    dw = torch.full((N, K), 5.0, device=dev)
    ops.linear_bwd_weight(dz, x, dw, None, beta=0.0)
    assert err(dw, dz.double().t() @ x.double()) < 2e-5


def test_crit_tail_equals_the_separate_launches(dev):
    """nacf_crit_tail_fwd / _bwd (the criterion's ONE tail launch each way) against nacf_nll_reduce_multi + nacf_kldiv_mean +
    nacf_loss_combine and their backward forms: the same bits in the slab, the total, the meters, gslab and the length head's gradient"""
    from nacf_amd.runtime import ops
    g = torch.Generator().manual_seed(5)
    rp, n_pass, S, B, Lm = 700, 2, 8, 16, 20
    ll = -torch.rand(n_pass * rp, generator=g).to(dev)
    labels = torch.randint(0, 9, (n_pass * rp,), generator=g).to(dev)          # 0 = <pad>, 4 = <mask> among them
    am = torch.where(torch.rand(n_pass * rp, generator=g).to(dev) < 0.5, labels, labels + 1)
    x = torch.log_softmax(torch.randn(B, Lm, generator=g), -1).to(dev)
    t = torch.softmax(torch.randn(B, Lm, generator=g), -1).to(dev)
    coef = torch.tensor([0.8 / B, 1.0 / B, 0.3], device=dev)
    m_dst = torch.tensor([0, 2, 3, 0, 4, 5, 6, 7, 1], dtype=torch.int32, device=dev)
    m_src = torch.tensor([0, 1, 2, S, S + 1, S + 2, S + 3, S + 4, 2 * S], dtype=torch.int32, device=dev)
    m_scale = torch.tensor([0.8, 1, 1, 1, 1, 1, 1, 1, float(B * Lm)], device=dev)
    # separate launches
    slab0, total0, meters0 = torch.zeros(3 * S, device=dev), torch.zeros(1, device=dev), torch.full((8,), 2.0, device=dev)
    ops.nll_reduce_multi(ll, am, labels, (True, False), [slab0[0:5], slab0[S:S + 5]])
    ops.kldiv_mean(x, t, slab0[2 * S:2 * S + 1], None)
    ops.loss_combine(slab0, 3, S, coef, total0, m_dst, m_src, m_scale, meters0)
    gt = torch.tensor([1.7], device=dev)
    gslab0, dx0 = torch.empty(3 * S, device=dev), torch.empty_like(x)
    ops.loss_combine_bwd(gt, coef, 3, S, gslab0)
    ops.kldiv_mean(x, t, None, dx0, gscale=gslab0[2 * S:2 * S + 1])
    # one launch each way
    tail = ops.CritTail()
    for i in range(n_pass):
        sl = slice(i * rp, (i + 1) * rp)
        tail.add_pass(ll[sl], am[sl], labels[sl], i == 0, i)
    tail.kl = (x, t, 2)
    slab1, total1, meters1 = torch.zeros(3 * S, device=dev), torch.zeros(1, device=dev), torch.full((8,), 2.0, device=dev)
    ops.crit_tail_fwd(tail, slab1, 3, S, coef, total1, m_dst, m_src, m_scale, meters1)
    gslab1, dx1 = torch.empty(3 * S, device=dev), torch.empty_like(x)
    ops.crit_tail_bwd(tail, gt, coef, 3, S, gslab1, dx1)
    assert torch.equal(slab0, slab1) and torch.equal(total0, total1) and torch.equal(meters0, meters1)
    assert torch.equal(gslab0, gslab1) and torch.equal(dx0, dx1)
    assert float(total1) != 0.0 and float(slab1[2 * S]) > 0.0


def test_rmsprop_step_equals_torch_rmsprop(dev):
    """nacf_rmsprop_step against clip_grad_value_ + torch.optim.RMSprop as misc/optim.py:52-60 constructs it (three steps, weight decay),
    and get_optimizer(opt['optim'] = 'rmsprop') drives it through the step engine's interface"""
    from nacf_amd.runtime import ops
    g = torch.Generator().manual_seed(3)
    n, lr, wd, clip = 5000, 3e-3, 5e-4, 0.5
    p0 = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) * 2 for _ in range(3)]
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.RMSprop([ref], lr=lr, weight_decay=wd)
    p, sq, lr_dev = p0.clone().to(dev), torch.zeros(n, device=dev), torch.full((1,), lr, device=dev)
    for gr in grads:
        ref.grad = gr.clone()
        torch.nn.utils.clip_grad_value_([ref], clip)
        opt.step()
        gd = gr.clone().to(dev)
        ops.rmsprop_step(p, gd, sq, lr_dev, 0.99, 1e-8, wd, clip, 1.0, zero_grad=True)
        assert float(gd.abs().max()) == 0.0
    assert float((p.cpu() - ref.detach()).abs().max()) < 2e-6
    import nacf_amd
    from nacf_amd.misc.optim import get_optimizer, FusedRMSprop
    o = nacf_amd.opts.make_opt("NAB", "MSRVTT", with_category=True, dim_hidden=64, num_attention_heads=4, intermediate_size=128, dim_i=32,
                               dim_m=32, max_len=10, vocab_size=101, optim="rmsprop")
    m = nacf_amd.get_model(o).to(dev)
    so = get_optimizer(m.opt, m)
    assert isinstance(so._optimizer, FusedRMSprop)
    w0 = m.flat.data.clone()
    m.flat.grad.fill_(0.1)
    so.step()
    assert float((m.flat.data - w0).abs().max()) > 0
