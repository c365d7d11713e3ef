"""GPU: the layer chain (csrc/gemm_bf16_chain.hpp; nacf_chain_begin / nacf_chain_flush behind ops.chain): the eight forward
launches of a decoder layer (models/bert.py:262-303: q|k|v, self-attention, output projection, cross-attention query,
cross-attention, output projection, FFN1, FFN2) as ONE persistent launch with device-wide barriers between its stages.

Bars: every stage runs the device function of its stand-alone kernel, so a chained layer is BIT-IDENTICAL to the same layer
launched call by call with the panel kernel forced on (NACF_GEMM_PANEL=1) -- outputs, loss, every gradient, dropout masks,
NA-decode tokens; against the default kernels (64x64 / 128x128 / wide) only the summation order of the GEMMs differs:
the fp32 kernels' tolerance.  The chain is OPT-IN (NACF_CHAIN=1): measured, it is slower than the call-by-call layer
(DESIGN.md section 4d); these tests keep it parity-green.  The model-width training test below also compares the chained layer
with the default kernels, which the rest of the suite pins to the reference fixtures and the oracle."""
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


def _model(opt, dev, seed=0):
    import nacf_amd
    from nacf_amd import synthetic as S
    from nacf_amd.runtime import ops
    ops.set_gemm_mode("bf16x3")
    m = nacf_amd.get_model(opt)
    m.load_state_dict(S.init_state_dict(opt, seed=seed))
    return m.to(dev)


def _last(L):
    return L.load().nacf_gemm_last_kernel().decode()


def test_chain_of_raw_calls_equals_call_by_call(dev, monkeypatch):
    """the C ABI itself: three dependent nn.Linear calls (row list, bias, activation, residual, dropout) around a self-attention
    core, queued between nacf_chain_begin and nacf_chain_flush, against the same calls launched one by one on the panel kernel
    (bit-identical) and on the default kernels (rounding); an ineligible call in the middle flushes and still computes the
    same values"""
    ops, L = _ops()
    from test_panel_gemm_gpu import Weights
    monkeypatch.setenv("NACF_GEMM_MODE", "bf16x3")
    D, H, R, Lq = 512, 8, 96, 20
    W = Weights(ops, dev, [(3 * D, D), (D, D), (1024, D), (200, D)], scale=0.05)
    try:
        g = torch.Generator().manual_seed(5)
        tok = (torch.rand(R, Lq, generator=g) < 0.6).long() * torch.randint(5, 90, (R, Lq), generator=g)
        tok[:, 0] = 7
        tok = tok.to(dev)
        rows = ops.rowset_build(tokens=tok.reshape(-1))
        x = rnd(R * Lq, D, seed=1).to(dev)
        b = [rnd(n, seed=20 + i, scale=0.1).to(dev) for i, n in enumerate((3 * D, D, 1024, 200))]
        rng = ops.RngState(77, dev)

        def run(chained, odd=False):
            qkv = torch.empty(R * Lq, 3 * D, device=dev)
            att = torch.empty(R * Lq, D, device=dev)
            a = torch.empty(R * Lq, D, device=dev)
            u = torch.empty(R * Lq, 1024, device=dev)
            v = torch.empty(R * Lq, 200, device=dev)
            kinds = []
            cm = ops.chain() if chained else None
            if cm is not None:
                cm.__enter__()
            ops.linear_fwd(x, W.w[0], qkv, ops.Epi(bias=b[0]), rows, zero_dead=True)
            kinds.append(_last(L))
            ops.attention_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], att, tok, 0, None, R, H, Lq, Lq, D // H, 1, R)
            if odd:          # N = 200: not a multiple of 128 -> flushes what is queued and runs behind it
                ops.linear_fwd(att, W.w[3], v, ops.Epi(bias=b[3]), rows, zero_dead=True)
                kinds.append(_last(L))
            ops.linear_fwd(att, W.w[1], a, ops.Epi(bias=b[1], p1=0.3, salt1=5, residual=x, row_tokens=tok.reshape(-1), rng=rng), rows, zero_dead=True)
            kinds.append(_last(L))
            ops.linear_fwd(a, W.w[2], u, ops.Epi(bias=b[2], act=L.ACT_BY_NAME["gelu_new"]), rows, zero_dead=True)
            kinds.append(_last(L))
            if cm is not None:
                cm.__exit__(None, None, None)
                kinds.append(_last(L))
            return (qkv, att, a, u, v), kinds

        monkeypatch.setenv("NACF_GEMM_PANEL", "1")
        ref, k_ref = run(False)
        assert all(k.startswith("gemm_panel_kernel") for k in k_ref), k_ref
        monkeypatch.setenv("NACF_CHAIN", "1")
        out, k_out = run(True)
        assert k_out[:3] == ["chain_queued"] * 3 and k_out[3].startswith("chain_kernel<2, 4>[4 stages: 3 linear, 1 attention]"), k_out
        for r_, o_ in zip(ref[:4], out[:4]):
            assert torch.equal(r_, o_)
        assert float(out[2].abs().max()) > 0 and bool(out[2].eq(0).any())
        # an ineligible call inside the block
        monkeypatch.setenv("NACF_CHAIN", "0")
        ref2, _ = run(False, odd=True)
        monkeypatch.setenv("NACF_CHAIN", "1")
        out2, k2 = run(True, odd=True)
        assert k2[0] == "chain_queued" and not k2[1].startswith("chain") and k2[-1].startswith("chain_kernel<2, 4>[2 stages: 2 linear, 0 attention]"), k2
        for r_, o_ in zip(ref2, out2):
            assert torch.equal(r_, o_)
        # the default kernels: another summation tree
        monkeypatch.setenv("NACF_GEMM_PANEL", "0")
        monkeypatch.setenv("NACF_CHAIN", "0")
        base, k_base = run(True)
        assert not any(k.startswith("chain") or k.startswith("gemm_panel") for k in k_base), k_base
        for r_, o_ in zip(base[:4], out[:4]):
            assert float((r_ - o_).abs().max()) < 2e-5 * max(1.0, float(r_.abs().max()))
        assert torch.equal(base[2].eq(0), out[2].eq(0))        # identical dropout masks / dead rows
        assert ops.chain_status() == 0
    finally:
        W.close()


@pytest.mark.parametrize("method", ["NACF", "NAB"])
def test_chained_layer_trains_bit_identically_to_the_call_by_call_panel_path(dev, monkeypatch, method):
    """a full-width training step (dropout 0.5): forward values, loss and EVERY gradient of the chained layer equal the
    call-by-call panel path bit for bit; against the default kernels they agree to rounding with identical dropout masks"""
    import nacf_amd
    from nacf_amd import synthetic as S
    from nacf_amd.misc.crit import get_criterion
    ops, L = _ops()
    opt = nacf_amd.opts.make_opt(method, "MSRVTT", with_category=True, max_len=20, vocab_size=1500, n_frames=12,
                                 fused_loss=True, hidden_dropout_prob=0.5, encoder_dropout=0.5)
    b = S.synth_batch(opt, 24, 12, seed=5)
    out = {}
    for tag, env in (("chain", dict(NACF_GEMM_PANEL="1", NACF_CHAIN="1")), ("calls", dict(NACF_GEMM_PANEL="1", NACF_CHAIN="0")),
                     ("default", dict(NACF_CHAIN="0", NACF_GEMM_PANEL="0"))):
        for k in ("NACF_GEMM_PANEL", "NACF_CHAIN"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        model = _model(opt, dev)
        model.train()
        crit = get_criterion(model.opt)
        model.zero_grad()
        two = method == "NACF"
        res = model(feats=[f.to(dev) for f in b["feats"]],
                    tgt_tokens=[b["tokens_1"].to(dev), b["tokens"].to(dev)] if two else b["tokens"].to(dev), category=b["category"].to(dev))
        res["tgt_word_labels"] = [b["labels_1"].to(dev), b["labels"].to(dev)] if two else b["labels"].to(dev)
        res["tgt_length"] = b["tgt_length"].to(dev)
        loss = crit.get_loss(res)
        loss.backward()
        out[tag] = (float(loss.detach()), model.flat.grad.clone())
    assert ops.chain_status() == 0
    (lc, gc), (lp, gp), (ld, gd) = out["chain"], out["calls"], out["default"]
    assert torch.isfinite(gc).all() and float(gc.abs().max()) > 0
    assert lc == lp and torch.equal(gc, gp)
    assert abs(lc - ld) < 1e-5 * abs(ld), (lc, ld)
    assert float((gc - gd).abs().max()) < 2e-4 * float(gd.abs().max())
    assert float(torch.nn.functional.cosine_similarity(gc, gd, dim=0)) > 0.999999


def test_chained_na_decode_returns_the_call_by_call_tokens(dev, monkeypatch):
    """mask-predict + coarse templates at model width, 6 length candidates (the query-subset last layer, the LDS-staged
    cross-attention over the length beam): chained passes return the tokens and scores of the call-by-call panel path, with
    and without the one-graph replay"""
    import nacf_amd
    from nacf_amd import synthetic as S
    from nacf_amd.models.Translator import Translator
    ops, L = _ops()
    opt = nacf_amd.opts.make_opt("NACF", "MSRVTT", with_category=True, max_len=20, vocab_size=1500, n_frames=12)
    b = S.synth_batch(opt, 16, 12, seed=7)
    feats, cat = [f.to(dev) for f in b["feats"]], b["category"].to(dev)
    res = {}
    for tag, env in (("chain", dict(NACF_GEMM_PANEL="1", NACF_CHAIN="1")), ("calls", dict(NACF_GEMM_PANEL="1", NACF_CHAIN="0"))):
        for k in ("NACF_GEMM_PANEL", "NACF_CHAIN"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        model = _model(opt, dev, seed=2)
        model.eval()
        for graph in ("off", "on"):
            dopt = dict(model.opt, paradigm="mp", use_ct=True, iterations=5, length_beam_size=6, beam_alpha=1.35, decode_graph=graph)
            with torch.no_grad():
                enc = model.encode(feats=feats)
                hyp, scores = Translator(model, dopt, device=dev).translate_batch(enc, cat, None, None)
            res[(tag, graph)] = (torch.as_tensor(hyp).clone(), scores)
    assert ops.chain_status() == 0
    def same(a, b):
        if isinstance(a, torch.Tensor):
            return torch.equal(a, b)
        if isinstance(a, (list, tuple)):
            return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
        return a == b
    for graph in ("off", "on"):
        assert torch.equal(res[("chain", graph)][0], res[("calls", graph)][0])
        assert same(res[("chain", graph)][1], res[("calls", graph)][1])
    assert torch.equal(res[("chain", "off")][0], res[("chain", "on")][0])
