"""GPU: attention_probs_dropout_prob > 0 (models/bert.py:135,169; reference default 0).  The mask comes from the device Philox
stream, so the oracle's torch dropout cannot reproduce it; instead the mask is READ BACK through the kernel itself (Q = 0 makes
the soft-max uniform, and the returned probabilities are the dropped ones) and a torch restatement with that mask checks the
forward and -- through autograd -- all three gradients.  Then a model with p > 0 trains under the graph engine."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    import nacf_amd  # noqa: F401
    from nacf_amd.runtime import ops
    return ops


@pytest.mark.parametrize("geom", ["self", "cross"])
def test_attention_probability_dropout_against_a_torch_restatement_with_the_same_mask(dev, geom):
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    H, dk, p, salt = 4, 16, 0.3, 77
    D = H * dk
    if geom == "self":
        R, Lq, Lk, kv_div, kv_mod, n_kv, causal = 6, 9, 9, 1, 6, 6, 1
        tokens = torch.randint(1, 50, (R, Lq), generator=g)
        tokens[:, 6:] = 0
        tokens[0, 3:] = 0
        tokens = tokens.to(dev)
    else:
        R, Lq, Lk, kv_div, kv_mod, n_kv, causal, tokens = 6, 7, 40, 2, 3, 3, 0, None
    rnd = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).to(dev)
    q, k, v = rnd(R * Lq, D), rnd(n_kv * Lk, D), rnd(n_kv * Lk, D)
    rng = ops.RngState(11, dev)
    drop = (p, salt, rng)

    def fwd(qq, want_probs):
        out = torch.empty(R * Lq, D, device=dev)
        probs = torch.empty(H, R, Lq, Lk, device=dev) if want_probs else None
        ops.attention_fwd(qq, k, v, out, tokens, causal, probs, R, H, Lq, Lk, dk, kv_div, kv_mod, drop=drop)
        return out, probs

    # the mask: uniform soft-max over the unmasked keys, dropped -> keep-scale = probs * (number of unmasked keys)
    _, pz = fwd(torch.zeros_like(q), True)
    kvrow = (torch.arange(R, device=dev) // kv_div) % kv_mod
    allowed = torch.ones(R, Lq, Lk, dtype=torch.bool, device=dev)
    if tokens is not None:
        allowed &= (tokens != 0).unsqueeze(1)
    if causal:
        allowed &= torch.tril(torch.ones(Lq, Lk, dtype=torch.bool, device=dev)).unsqueeze(0)
    n_ok = allowed.sum(-1, keepdim=True).clamp(min=1).float()
    mask = pz * n_ok.unsqueeze(0)                                    # [H, R, Lq, Lk]: 0 or 1 / (1 - p) where allowed
    live = allowed.unsqueeze(0).expand_as(mask)
    vals = mask[live]
    assert bool(((vals == 0) | ((vals - 1 / (1 - p)).abs() < 1e-4)).all())
    assert abs(float((vals > 0).float().mean()) - (1 - p)) < 0.05
    mask = torch.where(live, (mask > 0).float() / (1 - p), torch.zeros_like(mask))

    def restate(qq, kk, vv):
        qh = qq.view(R, Lq, H, dk).permute(2, 0, 1, 3)                # [H, R, Lq, dk]
        kh = kk.view(n_kv, Lk, H, dk)[kvrow].permute(2, 0, 1, 3)
        vh = vv.view(n_kv, Lk, H, dk)[kvrow].permute(2, 0, 1, 3)
        s = qh @ kh.transpose(-1, -2) / math.sqrt(dk)
        s = s.masked_fill(~allowed.unsqueeze(0), -10e6)
        pr = torch.softmax(s, -1) * mask
        return (pr @ vh).permute(1, 2, 0, 3).reshape(R * Lq, D), pr

    out, probs = fwd(q, True)
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    ref, ref_p = restate(qr, kr, vr)
    # (fully masked query rows -- <pad> queries of the self-attention -- are uniform over masked keys in both; compare live rows)
    rows_ok = allowed.any(-1).reshape(-1) if tokens is None else (tokens != 0).reshape(-1)
    assert float((out - ref.detach())[rows_ok].abs().max()) < 2e-5
    assert float((probs - ref_p.detach())[live].abs().max()) < 2e-6
    do = rnd(R * Lq, D) * rows_ok.unsqueeze(1)
    ref.backward(do)
    dq, dk_, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ops.attention_bwd(q, k, v, do, dq, dk_, dv, tokens, causal, R, n_kv, H, Lq, Lk, dk, kv_div, kv_mod, drop=drop)
    assert float((dq - qr.grad)[rows_ok].abs().max()) < 5e-5
    assert float((dk_ - kr.grad).abs().max()) < 5e-5 and float((dv - vr.grad).abs().max()) < 5e-5
    # p = 0 through the same entry: the kernels without dropout, bit for bit
    o0, o1 = torch.empty_like(out), torch.empty_like(out)
    ops.attention_fwd(q, k, v, o0, tokens, causal, None, R, H, Lq, Lk, dk, kv_div, kv_mod, drop=(0.0, salt, rng))
    ops.attention_fwd(q, k, v, o1, tokens, causal, None, R, H, Lq, Lk, dk, kv_div, kv_mod)
    assert torch.equal(o0, o1)


def test_a_model_with_attention_dropout_trains_under_the_graph_engine(dev):
    import nacf_amd
    from nacf_amd import synthetic as S
    from nacf_amd.misc.crit import get_criterion
    from nacf_amd.misc.optim import get_optimizer
    from nacf_amd.misc.run import get_forword_results
    from nacf_amd.runtime.engine import TrainStep
    opt = nacf_amd.opts.make_opt("NACF", "MSRVTT", with_category=True, max_len=20, vocab_size=500, n_frames=8, fused_loss=True,
                                 attention_probs_dropout_prob=0.1, learning_rate=2e-3)
    model = nacf_amd.get_model(opt)
    model.load_state_dict(S.init_state_dict(opt, seed=1))
    model.to(dev).train()
    crit, optim = get_criterion(model.opt), get_optimizer(model.opt, model)
    b = S.synth_batch(opt, 16, 8, seed=2)
    batch = {"feats": [f.to(dev) for f in b["feats"]], "tokens": b["tokens"].to(dev), "labels": b["labels"].to(dev),
             "category": b["category"].to(dev), "length_target": b["tgt_length"].to(dev),
             "tokens_1": b["tokens_1"].to(dev), "labels_1": b["labels_1"].to(dev)}
    engine = TrainStep(model, crit, optim, lambda bb: get_forword_results(model.opt, model, bb, dev), graph="on")
    engine(batch)
    first = float(engine.loss)
    for _ in range(40):
        engine()
    assert engine.captured and math.isfinite(float(engine.loss)) and float(engine.loss) < 0.9 * first, (first, float(engine.loss))
    # evaluation is deterministic and unaffected by the dropout sites
    model.eval()
    with torch.no_grad():
        e = model.encode(feats=batch["feats"])
        h1 = model.decoder(batch["tokens"], enc_output=e["enc_output"], category=batch["category"])[0]
        h2 = model.decoder(batch["tokens"], enc_output=e["enc_output"], category=batch["category"])[0]
    t = lambda x: x[-1] if isinstance(x, list) else x
    assert torch.equal(t(h1), t(h2))
