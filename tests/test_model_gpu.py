"""GPU: the drop-in surface (get_model / Seq2Seq / Criterion / optimiser /
Translator) on the HIP path vs (a) the golden vectors captured from the
reference and (b) the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): logits / log-probs within 1e-3 fp32 (we hold
5e-5 at the tiny shapes, 2e-4 at d=512), greedy NA-decode token ids bit-exact.
"""
import pytest
import torch

from oracle import nacf_oracle as O
from util import gold_batch, gold_json, gold_opt, gold_state, load_gold, maxerr, t

pytestmark = pytest.mark.gpu


def build(opt, sd, dev, **extra):
    import nacf_amd
    o = dict(opt)
    o.update(extra)
    m = nacf_amd.get_model(o)
    m.load_state_dict({k: v.clone() for k, v in sd.items()})
    return m.to(dev)


def _tok_labels(opt, b):
    vw = opt["visual_word_generation"]
    tokens = [b["tokens_1"], b["tokens"]] if vw else b["tokens"]
    labels = [b["labels_1"], b["labels"]] if vw else b["labels"]
    return tokens, labels


TRAIN_CASES = ["tiny_nacf_train", "tiny_nab_train", "tiny_arb2_train", "tiny_arb_train", "tiny_arb_watch_train", "tiny_nab_nogate_train", "tiny_nacf_pmlm_train", "tiny_nab_pmlm_ln_train", "tiny_nacf_lwe_train", "tiny_nab_variants_train",
               "tiny_nacf_ln_train", "tiny_nacf_pos_train", "tiny_nacf_pos_ln_train"]


def _force_tile(monkeypatch, tile):
    """tile None = the production heuristic (nothing forced); "64" / "128" pin the workgroup tile of every GEMM"""
    import os
    assert "NACF_GEMM_TILE" not in os.environ
    if tile is not None:
        monkeypatch.setenv("NACF_GEMM_TILE", tile)


@pytest.mark.parametrize("tile", [None, "64", "128"])
@pytest.mark.parametrize("name", TRAIN_CASES)
@pytest.mark.parametrize("fused", [False, True])
def test_train_step_vs_reference_golden(dev, name, fused, tile, monkeypatch):
    _force_tile(monkeypatch, tile)
    from nacf_amd.misc.crit import get_criterion
    from nacf_amd.misc.optim import get_optimizer
    g = load_gold(name)
    opt = gold_opt(g)
    b = gold_batch(g, dev)
    sd = O.init_state_dict(opt, seed=0)
    model = build(opt, sd, dev, fused_loss=fused)
    model.train()
    crit = get_criterion(model.opt)
    optim = get_optimizer(model.opt, model)
    tokens, labels = _tok_labels(opt, b)
    optim.zero_grad()
    res = model(feats=b["feats"], tgt_tokens=tokens, category=b["category"])
    assert maxerr(res["enc_output"], t(g["out.enc_output"])) < 2e-5
    if "out.pred_length" in g.files:
        assert maxerr(res["pred_length"], t(g["out.pred_length"])) < 2e-5
        res["tgt_length"] = b["tgt_length"]
    if not fused:
        for i, lp in enumerate(res["tgt_word_logprobs"]):
            assert lp.shape == tuple(g[f"out.logprobs{i}"].shape)
            assert maxerr(lp, t(g[f"out.logprobs{i}"])) < 5e-5
    res["tgt_word_labels"] = labels
    loss = crit.get_loss(res)
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * max(1.0, abs(float(g["loss"])))
    loss.backward()
    clip = opt["grad_clip"]
    worst = ("", 0.0)
    for k, p in model.named_parameters():
        ref = t(g["grad." + k])
        e = maxerr(p.grad.clamp(-clip, clip), ref)
        if e > worst[1]:
            worst = (k, e)
    assert worst[1] < 5e-5, worst
    optim.step()
    grads_ref = gold_state(g, "grad.")
    for k, v in gold_state(g, "after.").items():
        cur = model.state_dict()[k].detach().cpu()
        if v.is_floating_point():
            # Adam step 1 is lr*g/(|g|+eps): where the reference gradient itself is round-off noise
            # (|g| ~ eps, e.g. attention key biases, whose exact gradient is 0) the step direction is
            # arbitrary in the reference too -- there only the step SIZE (<= lr) is checked.
            d = (cur - v).abs()
            if k in grads_ref:
                solid = grads_ref[k].abs() > 2e-6
                assert float(d[solid].max() if solid.any() else 0.0) < 1.5e-4, k
                assert float(d.max()) <= 2.05 * opt["learning_rate"], k
            else:
                assert float(d.max()) < 1.5e-4, k
        else:
            assert int(cur) == int(v), k
    # meters: loss info names/values as the reference's Criterion reports them
    names, info = crit.get_loss_info()
    ref = dict(zip([str(n) for n in g["loss_names"]], g["loss_info"].tolist()))
    mine = dict(zip(names, info))
    for k, v in ref.items():
        assert abs(mine[k] - v) < 1e-3 * max(1.0, abs(v)), (k, mine[k], v)


def _run_decode(model, dec, b, dev, teacher=None, t_enc=None, graph="off", collect=True, **more):
    from nacf_amd.models.Translator import Translator
    dopt = dict(model.opt)
    dopt.update(dec)
    dopt.update(collect_best_candidate_iterative_results=collect, not_only_best_candidate=True, decode_graph=graph)
    dopt.update(more)
    tr = Translator(model, dopt, device=dev, teacher_model=teacher)
    with torch.no_grad():
        enc = model.encode(feats=b["feats"])
    gold = b["gold_tokens"].clone() if dec.get("load_generated_captions") else None      # the captions whose lengths seed the beam
    hyp, extra = tr.translate_batch(enc, b["category"], gold, None, teacher_encoder_outputs=t_enc)
    return enc, hyp, extra


@pytest.mark.parametrize("tile", [None, "64", "128"])
@pytest.mark.parametrize("graph", ["off", "on"])
@pytest.mark.parametrize("name", ["tiny_nacf_decode", "tiny_nab_decode", "tiny_nacf_goldlen_decode"])
def test_na_decode_tokens_bit_exact_vs_reference_golden(dev, name, graph, tile, monkeypatch):
    """graph='on': mask-predict variants replay from one hipGraph on a max_len-1 wide canvas and must still return the
    reference's tokens, per-iteration tokens / probabilities and shapes"""
    _force_tile(monkeypatch, tile)
    g = load_gold(name)
    opt = gold_opt(g)
    b = gold_batch(g, dev)
    model = build(opt, O.init_state_dict(opt, seed=3), dev)
    model.eval()
    variants = sorted({k.split(".")[0] for k in g.files if k.endswith(".hyp")})
    for v in variants:
        dec = gold_json(g, v + ".dec_json")
        enc, hyp, (it_tok, it_prob) = _run_decode(model, dec, b, dev, graph=graph)
        if graph == "on" and dec.get("paradigm", "mp") == "mp":
            assert len([k for k in model._nacf_decode_graphs if k[0] != "seen"]) >= 1
            # a second batch through the SAME captured graph: permuted videos give permuted captions
            perm = torch.arange(hyp.shape[0] - 1, -1, -1, device=dev)
            b2 = dict(b, feats=[f[perm] for f in b["feats"]], category=b["category"][perm])
            if "gold_tokens" in b:
                b2["gold_tokens"] = b["gold_tokens"][perm]
            _, hyp2, _ = _run_decode(model, dec, b2, dev, graph=graph)
            w = min(hyp.shape[1], hyp2.shape[1])
            assert torch.equal(hyp2[:, :w], hyp[perm][:, :w])
        assert maxerr(enc["enc_output"], t(g["out.enc_output"])) < 2e-5
        assert maxerr(enc["pred_length"], t(g["out.pred_length"])) < 2e-5
        assert torch.equal(hyp.cpu(), t(g[v + ".hyp"])), v
        assert torch.equal(it_tok.cpu(), t(g[v + ".iter_tokens"]).long()), v
        assert maxerr(it_prob, t(g[v + ".iter_probs"])) < 2e-5, v


@pytest.mark.parametrize("extra", [dict(load_word_embeddings=True), dict(parallel_mlm=True), dict(pos_attention=True),
                                   dict(with_layernorm=True), dict(pos_attention=True, with_layernorm=True),
                                   dict(num_hidden_layers_decoder=2, enhance_input=0)])
def test_na_decode_graph_replay_equals_launch_by_launch_for_the_option_variants(dev, extra):
    """mask-predict with coarse templates, easy-first and left-to-right of option-variant models: the one-graph replay returns
    what the launch-by-launch loop returns (tokens, per-iteration tokens; probabilities bit for bit)"""
    g = load_gold("tiny_nacf_decode")
    opt = dict(gold_opt(g), **extra)
    b = gold_batch(g, dev)
    model = build(opt, O.init_state_dict(opt, seed=3), dev)
    model.eval()
    for v in ("mp_ct", "mp", "ef_ct", "l2r_q2"):
        dec = gold_json(g, v + ".dec_json")
        outs = {}
        for graph in ("off", "on"):
            _, hyp, (it_tok, it_prob) = _run_decode(model, dec, b, dev, graph=graph)
            outs[graph] = (hyp, it_tok, it_prob)
        assert torch.equal(outs["on"][0], outs["off"][0]), v
        assert torch.equal(outs["on"][1], outs["off"][1]) and torch.equal(outs["on"][2], outs["off"][2]), v
        assert int(outs["on"][0].ne(0).sum()) > 0


@pytest.mark.parametrize("name", ["tiny_nacf_decode", "tiny_nacf_goldlen_decode"])
def test_l2r_and_easy_first_replay_from_one_graph_with_the_reference_tokens(dev, name):
    """decoding/algorithms.py:275-418 read a slot count on the host between passes; captured into a hipGraph (and under
    opt['decode_fixed_passes']) they run their pass-count upper bounds instead -- the extra passes select nothing -- and
    must return the reference's tokens (golden fixtures), a second batch through the same graph included"""
    g = load_gold(name)
    opt = gold_opt(g)
    b = gold_batch(g, dev)
    model = build(opt, O.init_state_dict(opt, seed=3), dev)
    model.eval()
    variants = [v for v in sorted({k.split(".")[0] for k in g.files if k.endswith(".hyp")})
                if gold_json(g, v + ".dec_json").get("paradigm", "mp") in ("l2r", "ef")]
    assert variants
    for v in variants:
        dec = gold_json(g, v + ".dec_json")
        want = t(g[v + ".hyp"])
        n_graphs = len([k for k in model.__dict__.get("_nacf_decode_graphs", {}) if k[0] != "seen"])
        _, hyp_fixed, _ = _run_decode(model, dec, b, dev, graph="off", collect=False, decode_fixed_passes=True)
        _, hyp_graph, _ = _run_decode(model, dec, b, dev, graph="on", collect=False)
        assert len([k for k in model._nacf_decode_graphs if k[0] != "seen"]) == n_graphs + 1, v      # it WAS captured
        assert torch.equal(hyp_fixed.cpu(), want) and torch.equal(hyp_graph.cpu(), want), v
        perm = torch.arange(hyp_graph.shape[0] - 1, -1, -1, device=dev)
        b2 = dict(b, feats=[f[perm] for f in b["feats"]], category=b["category"][perm])
        if "gold_tokens" in b:
            b2["gold_tokens"] = b["gold_tokens"][perm]
        _, hyp2, _ = _run_decode(model, dec, b2, dev, graph="on", collect=False)
        w = min(hyp_graph.shape[1], hyp2.shape[1])
        assert torch.equal(hyp2[:, :w], hyp_graph[perm][:, :w]), v


def test_na_decode_with_ar_teacher_rescoring(dev):
    g = load_gold("tiny_nacf_teacher")
    opt, t_opt = gold_opt(g), gold_opt(g, "teacher_opt_json")
    b = gold_batch(g, dev)
    model = build(opt, O.init_state_dict(opt, seed=3), dev); model.eval()
    teacher = build(t_opt, O.init_state_dict(t_opt, seed=7), dev); teacher.eval()
    with torch.no_grad():
        t_enc = teacher.encode(feats=b["feats"])
    for graph in ("off", "on"):
        for v in ("mp_ct", "mp_md"):
            dec = gold_json(g, v + ".dec_json")
            _, hyp, (it_tok, _) = _run_decode(model, dec, b, dev, teacher, t_enc, graph=graph)
            assert torch.equal(hyp.cpu(), t(g[v + ".hyp"])), (graph, v)
            assert torch.equal(it_tok.cpu(), t(g[v + ".iter_tokens"]).long()), (graph, v)


@pytest.mark.parametrize("name", ["tiny_arb2_beam", "tiny_arb_beam", "tiny_arb_beam_eos", "tiny_arb2_beam_eos", "tiny_arb_watch_beam"])
def test_ar_beam_search_vs_reference_golden(dev, name):
    """config-5 comparator: batched device beam search == the reference's per-instance Beam objects"""
    from nacf_amd.models.Translator import Translator
    g = load_gold(name)
    opt = gold_opt(g)
    b = gold_batch(g, dev)
    sd = O.init_state_dict(opt, seed=11)
    sd["tgt_word_prj.weight"][O.EOS] *= float(g["eos_boost"])
    model = build(opt, sd, dev); model.eval()
    dopt = dict(model.opt, beam_size=int(g["beam_size"]), beam_alpha=float(g["alpha"]), topk=int(g["topk"]))
    with torch.no_grad():
        enc = model.encode(feats=b["feats"])
    hyps, scores = Translator(model, dopt, device=dev).translate_batch(enc, b["category"], None, None)
    assert len(hyps) == int(g["B"])
    for i, h in enumerate(hyps):
        assert len(h) == int(g["n_best"][i]), (i, h)
        for j, x in enumerate(h):
            n = int(g["hyp_len"][i, j])
            assert x == g["hyp"][i, j, :n].tolist(), (i, j, x)
            assert abs(scores[i][j] - float(g["score"][i, j])) < 2e-4


def test_full_shape_logits_and_decode(dev):
    """d=512, F=60, V=10547: weights regenerated from the seeded generator."""
    g = load_gold("full_nacf")
    opt = gold_opt(g)
    sd = O.init_state_dict(opt, seed=int(g["seed_weights"]))
    batch = O.synth_batch(opt, int(g["B"]), int(g["F"]), seed=int(g["seed_batch"]))
    model = build(opt, sd, dev); model.eval()
    feats = [f.to(dev) for f in batch["feats"]]
    cat = batch["category"].to(dev)
    with torch.no_grad():
        enc = model.encode(feats=feats)
        hid, *_ = model.decoder(batch["tokens"].to(dev), enc_output=enc["enc_output"], category=cat)
        logp = model.vocab_logprobs(hid)
    assert maxerr(enc["enc_output"][:, ::7, ::5], t(g["enc_output_sample"])) < 2e-4
    assert maxerr(enc["pred_length"], t(g["pred_length"])) < 2e-4
    # sampled logits: compare differences of log-probs inside a row with differences of reference logits
    idx = t(g["logit_idx"])
    V = opt["vocab_size"]
    rows, cols = idx // V, idx % V
    lp = logp.reshape(-1, V).cpu()
    ref_vals = t(g["logit_val"])
    # logits = logp + lse(row): recover lse from one anchor per row (first sample seen in that row)
    anchor = {}
    for r, c, v in zip(rows.tolist(), cols.tolist(), ref_vals.tolist()):
        anchor.setdefault(r, (c, v))
    lse = torch.tensor([anchor[r][1] - float(lp[r, anchor[r][0]]) for r in rows.tolist()])
    assert float((lp[rows, cols] + lse - ref_vals).abs().max()) < 1e-3      # north_star: logits within 1e-3
    margin = t(g["logit_margin"]).reshape(-1)
    safe = margin > 1e-5
    # the comparison below must not be vacuous: every margin the fixture marks as a tie is a <pad> slot (hidden state zeroed
    # by non_pad_mask, models/bert.py:271-299: all logits equal), and at least 99 % of the real positions are kept
    nonpad = batch["tokens"].reshape(-1) != 0
    assert nonpad.numel() == safe.numel() and float(safe[nonpad].float().mean()) > 0.99, float(safe[nonpad].float().mean())
    assert not bool(safe[~nonpad].any())
    am = lp.argmax(-1)
    assert torch.equal(am[safe], t(g["logit_argmax"]).reshape(-1).long()[safe])
    from nacf_amd.models.Translator import Translator
    dec = dict(paradigm="mp", use_ct=True, iterations=5, length_beam_size=6, beam_alpha=1.35)
    dopt = dict(model.opt); dopt.update(dec)
    dopt.update(collect_best_candidate_iterative_results=True, not_only_best_candidate=True)
    hyp, (it_tok, it_prob) = Translator(model, dopt, device=dev).translate_batch(enc, cat, None, None)
    assert torch.equal(hyp.cpu(), t(g["mp_ct.hyp"]))                          # bit-exact greedy NA tokens
    assert torch.equal(it_tok.cpu(), t(g["mp_ct.iter_tokens"]).long())
    assert maxerr(it_prob, t(g["mp_ct.iter_probs"])) < 1e-4


def test_reference_style_loop_with_torch_nllloss(dev):
    """the reference's own calling sequence (misc/run.py:254-261) with a plain
    torch criterion on the returned log-probs: gradients still land in the flat buffer"""
    g = load_gold("tiny_nab_train")
    opt = gold_opt(g)
    b = gold_batch(g, dev)
    model = build(opt, O.init_state_dict(opt, seed=0), dev); model.train()
    res = model(feats=b["feats"], tgt_tokens=b["tokens"], category=b["category"], opt=opt, vocab=None)
    lp = res["tgt_word_logprobs"][0]
    lab = b["labels"]
    loss = torch.nn.functional.nll_loss(lp.reshape(-1, lp.shape[-1]), lab.reshape(-1), ignore_index=0,
                                        reduction="sum") / lp.shape[0]
    kl = torch.nn.functional.kl_div(res["pred_length"], b["tgt_length"], reduction="mean")
    (loss + kl).backward()
    torch.nn.utils.clip_grad_value_(model.parameters(), opt["grad_clip"])
    for k, p in model.named_parameters():
        assert maxerr(p.grad, t(g["grad." + k])) < 5e-5, k
    assert model.flat.grads_attached()


def test_padding_invariance_and_permutation_equivariance(dev):
    g = load_gold("tiny_nacf_decode")
    opt = gold_opt(g)
    b = gold_batch(g, dev)
    model = build(opt, O.init_state_dict(opt, seed=3), dev); model.eval()
    with torch.no_grad():
        enc = model.encode(feats=b["feats"])
        B = enc["enc_output"].shape[0]
        tok = torch.tensor([[7, 8, 9, 10, 0, 0], [11, 12, 13, 14, 15, 16], [6, 6, 6, 0, 0, 0], [9, 8, 7, 6, 5, 0]],
                           device=dev)[:B]
        h, *_ = model.decoder(tok, enc_output=enc["enc_output"], category=b["category"])
        assert float(h[tok.eq(0)].abs().max()) == 0.0                  # PAD rows exactly zero (bert.py:271-299)
        wide = torch.cat([tok, torch.zeros(B, 3, dtype=torch.long, device=dev)], 1)
        h2, *_ = model.decoder(wide, enc_output=enc["enc_output"], category=b["category"])
        assert maxerr(h2[:, :6], h) < 1e-6                             # extra PAD columns change nothing
        perm = torch.tensor([2, 0, 3, 1], device=dev)[:B]
        h3, *_ = model.decoder(tok[perm], enc_output=enc["enc_output"][perm], category=b["category"][perm])
        assert maxerr(h3, h[perm]) < 1e-6


def test_dropout_training_step_is_finite_and_replayable(dev):
    """p=0.5 dropout (reference default): masks are regenerated in backward from
    (seed, step): two models with the same seed produce identical gradients."""
    opt = gold_opt(load_gold("tiny_nacf_train"))
    opt.update(hidden_dropout_prob=0.5, encoder_dropout=0.5, seed=11)
    b = O.synth_batch(opt, 6, 6, seed=2)
    grads = []
    for _ in range(2):
        model = build(opt, O.init_state_dict(opt, seed=0), dev, fused_loss=True); model.train()
        from nacf_amd.misc.crit import get_criterion
        crit = get_criterion(model.opt)
        res = model(feats=[f.to(dev) for f in b["feats"]], tgt_tokens=[b["tokens_1"].to(dev), b["tokens"].to(dev)],
                    category=b["category"].to(dev))
        res["tgt_word_labels"] = [b["labels_1"].to(dev), b["labels"].to(dev)]
        res["tgt_length"] = b["tgt_length"].to(dev)
        loss = crit.get_loss(res)
        loss.backward()
        assert torch.isfinite(loss) and torch.isfinite(model.flat.grad).all()
        grads.append(model.flat.grad.clone())
    assert torch.equal(grads[0], grads[1])


def test_staged_backward_matches_single_pass(dev):
    """runtime/ddp.py: backward split at the encoder outputs (the overlap point of the bucketed all-reduce)
    leaves the same gradients in the flat buffer as loss.backward(), and the flat layout separates the buckets"""
    import nacf_amd
    from nacf_amd.misc.crit import get_criterion
    from nacf_amd.runtime.ddp import DataParallel
    from nacf_amd import synthetic as S
    g = load_gold("tiny_nacf_train")
    opt = dict(gold_opt(g), fused_loss=True)
    b = gold_batch(g)
    grads = []
    for staged in (False, True):
        model = nacf_amd.get_model(opt)
        model.load_state_dict(S.init_state_dict(opt, seed=0))
        model.to(dev).train()
        crit = get_criterion(model.opt)
        ddp = DataParallel(model)
        model.zero_grad()
        res = model(feats=[f.to(dev) for f in b["feats"]], tgt_tokens=[b["tokens_1"].to(dev), b["tokens"].to(dev)],
                    category=b["category"].to(dev))
        res["tgt_word_labels"] = [b["labels_1"].to(dev), b["labels"].to(dev)]
        res["tgt_length"] = b["tgt_length"].to(dev)
        loss = crit.get_loss(res)
        if staged:
            split = ddp.bucket_split()
            assert split is not None and 0 < split < model.flat.total
            cut, gr = ddp.backward_to_cut(loss)
            late = model.flat.grad[split:].clone()
            early_before = float(model.flat.grad[:split].abs().max())
            ddp.backward_from_cut(cut, gr)
            assert early_before == 0.0                                    # stage 1 left the encoder side untouched
            assert torch.equal(late, model.flat.grad[split:])             # stage 2 left the decoder side untouched
        else:
            loss.backward()
        grads.append(model.flat.grad.clone())
    assert float(grads[0].abs().max()) > 0
    assert float((grads[0] - grads[1]).abs().max()) <= 1e-6 * float(grads[0].abs().max())


def test_live_row_path_equals_dense_path_at_model_width(dev):
    """size-independent property of the <pad>-skipping GEMMs: at the real model width (d=512, V=10547, F=60) the step with
    live-row lists (default) and the step that multiplies every slot (pack_rows=False) give the same loss and the same
    gradients -- the skipped rows only ever held exact zeros"""
    import nacf_amd
    from nacf_amd.misc.crit import get_criterion
    from nacf_amd import synthetic as S
    outs = []
    for pack in (True, False):
        opt = nacf_amd.opts.make_opt("NACF", "MSRVTT", with_category=True, max_len=20, vocab_size=10547, fused_loss=True,
                                     hidden_dropout_prob=0.0, encoder_dropout=0.0, pack_rows=pack)
        b = S.synth_batch(opt, 16, 60, seed=4)
        model = nacf_amd.get_model(opt)
        model.load_state_dict(S.init_state_dict(opt, seed=1))
        model.to(dev).train()
        crit = get_criterion(model.opt)
        model.zero_grad()
        res = model(feats=[f.to(dev) for f in b["feats"]], tgt_tokens=[b["tokens_1"].to(dev), b["tokens"].to(dev)],
                    category=b["category"].to(dev))
        res["tgt_word_labels"] = [b["labels_1"].to(dev), b["labels"].to(dev)]
        res["tgt_length"] = b["tgt_length"].to(dev)
        loss = crit.get_loss(res)
        loss.backward()
        outs.append((float(loss.detach()), model.flat.grad.clone()))
    (l0, g0), (l1, g1) = outs
    assert abs(l0 - l1) <= 1e-5 * abs(l1)
    scale = float(g1.abs().max())
    assert scale > 0 and float((g0 - g1).abs().max()) <= 2e-5 * scale


@pytest.mark.parametrize("B,L", [(16, 20), (128, 20), (16, 30)])
def test_full_shape_train_step_vs_oracle(dev, B, L):
    """Training parity AT MODEL WIDTH (the shapes bench.py times: NACF, d=512, F=60, V=10547, B=128 is the bench batch;
    L=30 is the reference's MSRVTT default, opts.py:161-169): one step of the production kernels (heuristic tiles, live
    rows, fused loss, split-K) vs the CPU oracle's forward / criterion / backward / clip / Adam on this box's host
    cores (models/seq2seq.py:86-108, misc/crit.py:40-127, misc/run.py:254-261).  Dropout 0 (RNG streams cannot match).
    Bars: loss 1e-5 rel; every parameter gradient within 1e-4 of that tensor's max |g|; post-Adam weights within
    1.5e-4 where the gradient is above round-off (step 1 of Adam is lr*g/(|g|+eps): direction undefined on noise).
    The oracle is also evaluated in double, see below."""
    import os
    import nacf_amd
    from nacf_amd import synthetic as S
    from nacf_amd.misc.crit import get_criterion
    from nacf_amd.misc.optim import get_optimizer
    assert not [k for k in os.environ if k.startswith("NACF_GEMM_")]      # the production heuristic, nothing forced
    opt = nacf_amd.opts.make_opt("NACF", "MSRVTT", with_category=True, max_len=L, vocab_size=10547, n_frames=60,
                                 fused_loss=True, hidden_dropout_prob=0.0, encoder_dropout=0.0, beta=[0.35, 0.9])
    sd = S.init_state_dict(opt, seed=0)
    b = S.synth_batch(opt, B, 60, seed=11)
    model = build(opt, sd, dev)
    model.train()
    crit, optim = get_criterion(model.opt), get_optimizer(model.opt, model)
    optim.zero_grad()
    res = model(feats=[f.to(dev) for f in b["feats"]], tgt_tokens=[b["tokens_1"].to(dev), b["tokens"].to(dev)],
                category=b["category"].to(dev))
    res["tgt_word_labels"] = [b["labels_1"].to(dev), b["labels"].to(dev)]
    res["tgt_length"] = b["tgt_length"].to(dev)
    loss = crit.get_loss(res)
    loss.backward()
    grads = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()}
    optim.step()
    torch.cuda.synchronize()
    def oracle(dt):
        s_ = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        l_, _, g_ = O.train_step(s_, opt, [f.to(dt) for f in b["feats"]], [b["tokens_1"], b["tokens"]], b["category"],
                                 [b["labels_1"], b["labels"]], b["tgt_length"].to(dt), {}, lr=opt["learning_rate"])
        return float(l_), g_, s_
    # the oracle twice: as the reference runs it (fp32 on the CPU) and in double (the same restatement without
    # round-off: what both fp32 paths approximate).  Two fp32 evaluations with different summation orders differ by
    # ~1e-4 of a tensor's max on gradients that are sums of thousands of cancelling terms (bias gradients over the
    # B*F = 7680 encoder rows), so the bar is: within 1e-4 of the double result, flat -- and therefore within
    # 1e-4 + (the CPU fp32 path's own distance from the double result) of the fp32 oracle.
    o_loss, o_grads, sd_o = oracle(torch.float32)
    d_loss, d_grads, _ = oracle(torch.float64)
    assert abs(float(loss) - o_loss) <= 1e-5 * abs(o_loss), (float(loss), o_loss)
    assert abs(float(loss) - d_loss) <= 1e-5 * abs(d_loss), (float(loss), d_loss)
    clip = opt["grad_clip"]
    table = []
    for k, g in grads.items():
        ref, ref64 = o_grads[k], d_grads[k]
        scale = float(ref64.abs().max())
        if scale <= 1e-7:          # exact-zero gradients (attention key biases): pure round-off in every evaluation
            assert float(g.abs().max()) < 1e-5, k
            continue
        e32 = float((g - ref).abs().max()) / scale
        e64 = float((g.double() - ref64).abs().max()) / scale
        cpu64 = float((ref.double() - ref64).abs().max()) / scale
        table.append((e64, e32, cpu64, scale, k))
    table.sort(reverse=True)
    for e64, e32, cpu64, scale, k in table[:6]:
        print("  %-58s max|g| %.2e   HIP-vs-double %.2e   HIP-vs-fp32-oracle %.2e   fp32-oracle-vs-double %.2e" % (k, scale, e64, e32, cpu64))
    worst = (table[0][4], table[0][0])
    for e64, e32, cpu64, scale, k in table:
        # within 1e-4 of the double result -- or, on the few cancellation-dominated sums where the reference's OWN fp32
        # evaluation is further off than that (the four encoder bias gradients at B = 128: sums over 7680 rows, CPU
        # fp32 1.0-1.3e-4 from double, both HIP modes 1.3-2.1e-4), within three times the reference's fp32 error
        assert e64 <= max(1e-4, 3.0 * cpu64), (k, e64, cpu64, scale)
        assert e32 <= max(1e-4, 3.0 * cpu64) + cpu64, (k, e32, e64, cpu64)      # triangle inequality through the double result
    # post-Adam weights.  Step 1 of Adam moves every weight by lr * g / (|g| + eps): where the gradient is known to a
    # few per cent the step is determined, elsewhere only its size (<= lr) is.  "solid" = gradient above round-off
    # AND at least 20x the distance between the two evaluations; it must cover most of every GEMM weight.
    new = model.state_dict()
    for k, ref in o_grads.items():
        d = (new[k].detach().cpu() - sd_o[k]).abs()
        # what Adam normalises is clip(g) + weight_decay * w (misc/optim.py:61-62): that sum must be well determined
        g64 = d_grads[k]
        eff = g64.clamp(-clip, clip) + opt["weight_decay"] * sd[k].double()
        noise = torch.maximum((grads[k].double() - g64).abs(), (ref.double() - g64).abs())
        solid = (eff.abs() > 2e-6) & (eff.abs() > 20 * noise)
        assert float(d[solid].max() if solid.any() else 0.0) < 1.5e-4, k
        assert float(d.max()) <= 2.05 * opt["learning_rate"], k
        if ref.dim() == 2 and ref.numel() >= 512 * 512 and "embeddings" not in k and "tgt_word_prj" not in k:
            assert float(solid.float().mean()) > 0.5, (k, float(solid.float().mean()))
    for k in ("joint_representation_learner.bn0.running_mean", "joint_representation_learner.bn1.running_var"):
        assert maxerr(new[k], sd_o[k]) < 1e-4, k
    print("full-shape train parity B=%d L=%d: loss %.6f, worst gradient error vs the double oracle %.2e of max (%s)"
          % (B, L, float(loss), worst[1], worst[0]))


@pytest.mark.parametrize("layers", [1, 2])
def test_decoder_query_subset_is_bit_identical_on_the_kept_slots(dev, layers):
    """NA decoding reads the decoder's hidden state only at the re-masked slots: with `out_row_set` the last layer
    computes just those rows (keys/values still span every live slot) and they equal the full pass bit for bit"""
    import nacf_amd
    from nacf_amd import synthetic as S
    from nacf_amd.runtime import ops
    opt = nacf_amd.opts.make_opt("NACF", "MSRVTT", with_category=True, max_len=20, vocab_size=1000, n_frames=8,
                                 num_hidden_layers_decoder=layers)
    model = nacf_amd.get_model(opt)
    model.load_state_dict(S.init_state_dict(opt, seed=2))
    model.to(dev).eval()
    b = S.synth_batch(opt, 6, 8, seed=3)
    feats, cat = [f.to(dev) for f in b["feats"]], b["category"].to(dev)
    tokens = b["tokens"].to(dev)
    g = torch.Generator().manual_seed(0)
    keep = ((torch.rand(tokens.shape, generator=g) < 0.4).to(dev) & (tokens != 0))
    with torch.no_grad():
        enc = model.encode(feats=feats)["enc_output"]
        full = model.decoder(tokens, enc_output=enc, category=cat)[0]
        rows = ops.rowset_build(tokens=tokens.reshape(-1), flags=keep.reshape(-1).to(torch.uint8))
        out = model.decoder(tokens, enc_output=enc, category=cat, out_row_set=rows)
    full = full[-1] if isinstance(full, list) else full
    sub = out[0][-1] if isinstance(out[0], list) else out[0]
    assert out[1] is None and int(keep.sum()) > 10
    assert torch.equal(sub[keep], full[keep])
    assert float(sub[~keep].abs().max()) == 0.0           # everything else is zero-filled, never garbage


def test_decoder_signals_join_the_additional_features(dev):
    """models/Decoder.py:141-142: `signals` are added to the embedding's additional features (or stand in for them);
    one row per video here, as the embedding kernel broadcasts one additional row over a video's slots"""
    import nacf_amd
    from nacf_amd import synthetic as S
    opt = nacf_amd.opts.make_opt("NACF", "MSRVTT", with_category=True, max_len=20, vocab_size=1000, n_frames=8)
    model = nacf_amd.get_model(opt)
    model.load_state_dict(S.init_state_dict(opt, seed=2))
    model.to(dev).eval()
    b = S.synth_batch(opt, 6, 8, seed=3)
    feats, cat, tokens = [f.to(dev) for f in b["feats"]], b["category"].to(dev), b["tokens"].to(dev)
    with torch.no_grad():
        enc = model.encode(feats=feats)["enc_output"]
        pooled = enc.mean(1)
        sig = torch.randn(pooled.shape, generator=torch.Generator().manual_seed(5)).to(dev) * 0.3
        take = lambda o: (o[0][-1] if isinstance(o[0], list) else o[0])
        base = take(model.decoder(tokens, enc_output=enc, category=cat, pooled_memory=pooled + sig))
        with_sig = take(model.decoder(tokens, enc_output=enc, category=cat, pooled_memory=pooled, signals=sig.unsqueeze(1)))
        plain = take(model.decoder(tokens, enc_output=enc, category=cat, pooled_memory=pooled))
        zero = take(model.decoder(tokens, enc_output=enc, category=cat, pooled_memory=pooled, signals=torch.zeros_like(sig)))
    assert torch.equal(with_sig, base) and torch.equal(zero, plain)
    assert float((with_sig - plain).abs().max()) > 1e-3
    with pytest.raises(NotImplementedError):
        model.decoder(tokens, enc_output=enc, category=cat, signals=torch.zeros(tokens.shape[0], tokens.shape[1], pooled.shape[1], device=dev))


@pytest.mark.parametrize("B", [1, 64])
def test_reference_default_shapes_train_and_decode(dev, B):
    """the reference's own MSRVTT defaults (opts.py: max_len 30, n_frames 8 -> a 16-slot visual memory, batch 64) and
    the batch-of-one latency setting (run.py:139-143): a training step is finite and moves every parameter group,
    graph-replayed decoding equals launch-by-launch decoding, and hypotheses are well formed"""
    import nacf_amd
    from nacf_amd import synthetic as S
    from nacf_amd.misc.crit import get_criterion
    from nacf_amd.misc.optim import get_optimizer
    from nacf_amd.models.Translator import Translator
    opt = nacf_amd.opts.make_opt("NACF", "MSRVTT", default=True, vocab_size=10547, n_frames=8, fused_loss=True)
    assert opt["max_len"] == 30 and opt["with_category"] and opt["length_beam_size"] == 6 and opt["use_ct"]
    model = nacf_amd.get_model(opt)
    model.load_state_dict(S.init_state_dict(opt, seed=0))
    model.to(dev).train()
    b = S.synth_batch(opt, B, 8, seed=5)
    feats, cat = [f.to(dev) for f in b["feats"]], b["category"].to(dev)
    crit, optim = get_criterion(model.opt), get_optimizer(model.opt, model)
    before = model.flat.data.clone()
    optim.zero_grad()
    res = model(feats=feats, tgt_tokens=[b["tokens_1"].to(dev), b["tokens"].to(dev)], category=cat)
    res["tgt_word_labels"] = [b["labels_1"].to(dev), b["labels"].to(dev)]
    res["tgt_length"] = b["tgt_length"].to(dev)
    loss = crit.get_loss(res)
    loss.backward()
    optim.step()
    assert torch.isfinite(loss) and bool(torch.isfinite(model.flat.data).all())
    assert float((model.flat.data - before).abs().max()) > 0
    model.eval()
    outs = {}
    for graph in ("off", "on"):
        tr = Translator(model, dict(model.opt, decode_graph=graph), device=dev)
        with torch.no_grad():
            hyp, _ = tr.translate_batch(model.encode(feats=feats), cat, None, None)
        outs[graph] = hyp
    assert torch.equal(outs["on"], outs["off"])
    hyp = outs["on"]
    assert hyp.shape[0] == B and 4 <= hyp.shape[1] <= 29
    assert int(hyp.min()) >= 0 and int(hyp.max()) < 10547 and not bool((hyp == 4).any())     # no <mask> left


def test_adam_in_two_parts_equals_one_launch(dev):
    """data-parallel steps update the decoder-side parameters while the encoder-side gradient bucket is still being
    reduced: Adam over flat[split:] (bumping the step) then flat[:split] must equal one launch over everything"""
    import nacf_amd
    from nacf_amd import synthetic as S
    from nacf_amd.misc.optim import get_optimizer
    from nacf_amd.runtime.ddp import DataParallel
    opt = nacf_amd.opts.make_opt("NACF", "MSRVTT", with_category=True, dim_hidden=64, num_attention_heads=4,
                                 intermediate_size=128, dim_i=32, dim_m=32, max_len=10, vocab_size=101, n_frames=6)
    outs = []
    for parts in (False, True):
        model = nacf_amd.get_model(opt)
        model.load_state_dict(S.init_state_dict(opt, seed=0))
        model.to(dev)
        optim = get_optimizer(model.opt, model)
        split = DataParallel(model).bucket_split()
        assert 0 < split < model.flat.data.numel()
        g = torch.Generator().manual_seed(1)
        for step in range(3):
            model.flat.grad.copy_((torch.rand(model.flat.grad.numel(), generator=g) * 12 - 6).to(dev))   # some beyond the clip
            optim.step_update_learning_rate()
            if parts:
                optim._optimizer.step(grad_scale=0.5, lo=split, hi=None, bump=True)
                optim._optimizer.step(grad_scale=0.5, lo=0, hi=split, bump=False)
            else:
                optim._optimizer.step(grad_scale=0.5)
        outs.append((model.flat.data.clone(), int(optim._optimizer.step_dev)))
    assert outs[0][1] == outs[1][1] == 3
    assert torch.equal(outs[0][0], outs[1][0])


@pytest.mark.parametrize("name", ["tiny_nacf_train", "tiny_nab_train"])
def test_three_stage_backward_matches_single_pass(dev, name):
    """runtime/ddp.py: backward in three stages (loss -> decoder output -> encoder outputs -> inputs), the boundaries of
    the three gradient buckets: each stage completes exactly its own bucket and the total equals loss.backward()"""
    import nacf_amd
    from nacf_amd.misc.crit import get_criterion
    from nacf_amd.runtime.ddp import DataParallel
    from nacf_amd import synthetic as S
    g = load_gold(name)
    opt = dict(gold_opt(g), fused_loss=True)
    b = gold_batch(g, dev)
    tokens, labels = _tok_labels(opt, b)
    grads = []
    for staged in (False, True):
        model = nacf_amd.get_model(opt)
        model.load_state_dict(S.init_state_dict(opt, seed=0))
        model.to(dev).train()
        crit = get_criterion(model.opt)
        ddp = DataParallel(model)
        model.zero_grad()
        res = model(feats=b["feats"], tgt_tokens=tokens, category=b["category"])
        res["tgt_word_labels"] = labels
        res["tgt_length"] = b["tgt_length"]
        loss = crit.get_loss(res)
        if staged:
            s1, s2 = ddp.bucket_split(), ddp.head_split()
            assert s1 is not None and s2 is not None and 0 < s1 < s2 < model.flat.total
            assert model.flat.total - s2 == sum(p.numel() for p in model.tgt_word_prj.parameters())
            G = model.flat.grad
            hc, hg = ddp.backward_head(loss)
            assert float(G[:s2].abs().max()) == 0.0 and float(G[s2:].abs().max()) > 0       # only the vocabulary projection
            head = G[s2:].clone()
            cut, gr = ddp.backward_mid(hc, hg)
            assert float(G[:s1].abs().max()) == 0.0 and float(G[s1:s2].abs().max()) > 0 and torch.equal(head, G[s2:])
            mid = G[s1:s2].clone()
            ddp.backward_from_cut(cut, gr)
            assert torch.equal(head, G[s2:]) and torch.equal(mid, G[s1:s2]) and float(G[:s1].abs().max()) > 0
        else:
            loss.backward()
        grads.append(model.flat.grad.clone())
    assert float((grads[0] - grads[1]).abs().max()) <= 1e-6 * float(grads[0].abs().max())


def test_tied_weights_fall_back_to_two_buckets(dev):
    import nacf_amd
    from nacf_amd.runtime.ddp import DataParallel
    g = load_gold("tiny_nab_variants_train")
    opt = dict(gold_opt(g), fused_loss=True)
    assert opt["tie_weights"]
    model = nacf_amd.get_model(opt).to(dev)
    assert DataParallel(model).head_split() is None


def test_memory_fanout_accumulates_in_place_only_into_a_buffer_it_owns(dev):
    """MemoryFanoutFn.backward adds the time-mean's gradient INTO the incoming gradient of the visual memory
    (one read-modify-write instead of a second [B, T, D] tensor and an add).  That buffer is the K|V projection's freshly
    allocated dX, which nobody else holds: a retain_grad() on enc_output keeps its own clone (the gradient of enc_output as a
    decoder input, WITHOUT the pooled paths), and the encoder still receives the sum.  The contract for user hooks is the usual
    one: a hook that stashes its argument without cloning sees later in-place updates."""
    import nacf_amd
    from nacf_amd import synthetic as S
    from nacf_amd.misc.crit import get_criterion
    opt = nacf_amd.opts.make_opt("NACF", "MSRVTT", with_category=True, max_len=12, vocab_size=300, n_frames=6, dim_hidden=128,
                                 num_attention_heads=4, intermediate_size=256, dim_i=64, dim_m=64, fused_loss=True,
                                 hidden_dropout_prob=0.0, encoder_dropout=0.0)
    sd = S.init_state_dict(opt, seed=0)
    b = S.synth_batch(opt, 8, 6, seed=3)
    grads = {}
    for mode in ("plain", "retain"):
        model = build(opt, sd, dev)
        model.train()
        crit = get_criterion(model.opt)
        model.zero_grad()
        res = model(feats=[f.to(dev) for f in b["feats"]], tgt_tokens=[b["tokens_1"].to(dev), b["tokens"].to(dev)],
                    category=b["category"].to(dev))
        res["tgt_word_labels"] = [b["labels_1"].to(dev), b["labels"].to(dev)]
        res["tgt_length"] = b["tgt_length"].to(dev)
        stash = {}
        if mode == "retain":
            eo = res["enc_output"]
            eo.retain_grad()
            eo.register_hook(lambda g_: stash.__setitem__("clone", g_.clone()) or None)
        crit.get_loss(res).backward()
        grads[mode] = model.flat.grad.clone()
        if mode == "retain":
            assert torch.equal(res["enc_output"].grad, stash["clone"])          # the retained gradient was not mutated
            assert float(stash["clone"].abs().max()) > 0
    assert torch.equal(grads["plain"], grads["retain"])                         # and the parameters' gradients are the same


def test_automatic_mask_yields_the_reference_entry(dev):
    """opt['automatic_mask'] (models/seq2seq.py:37-42): results['attention_mask'] = [feat.sum(-1).eq(0) per modality]; the rest of the
    encode is unchanged (nothing reads the entry)"""
    g = load_gold("tiny_nacf_decode")
    opt = gold_opt(g)
    b = gold_batch(g, dev)
    feats = [f.clone() for f in b["feats"]]
    feats[0][1, 2:] = 0.0                                                  # zero-padded frames of one clip
    plain = build(opt, O.init_state_dict(opt, seed=3), dev); plain.eval()
    masked = build(opt, O.init_state_dict(opt, seed=3), dev, automatic_mask=True); masked.eval()
    with torch.no_grad():
        e0, e1 = plain.encode(feats=feats), masked.encode(feats=feats)
    assert "attention_mask" not in e0 and "attention_mask" in e1
    am = e1["attention_mask"]
    assert len(am) == len(feats) and all(torch.equal(a, f.sum(-1).eq(0)) for a, f in zip(am, feats))
    assert bool(am[0][1, 2:].all()) and not bool(am[0][0].any())
    assert torch.equal(e0["enc_output"], e1["enc_output"])
