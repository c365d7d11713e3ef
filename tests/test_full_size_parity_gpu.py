"""GPU: parity AT MODEL WIDTH for the paths bench.py times but the golden fixtures only cover at fixture size
("close the full-size parity holes"):

  (a) NACF mask-predict + coarse templates decode, d=512 / F=60 / V=10547, B = 128 (bench batch) and 32, hipGraph on and
      off, vs oracle.generate on this box's host cores (decoding/na_generate.py:14-135, decoding/algorithms.py:136-273):
      per-iteration tokens of every length candidate and the final hypotheses BIT-EXACT wherever the oracle's own decision
      was not a numerical tie (top-1/top-2 logit margin, the k-th / (k+1)-th confidence of select_worst, the best / second
      candidate score -- all recorded while the oracle runs); the tie-free share must be >= 0.99.
  (b) ARB2 beam-5 (models/Beam.py:5-169, models/Translator.py:24-161) at d=512 / V=10547, B=32: hypotheses + scores.
  (c) BASELINE configs[0] at its real size: NAB, Youtube2Text-shape (no category, beta=[0,1]), B=16, 60x2048 features:
      one training step vs the oracle (loss, every gradient).
"""
import pytest
import torch

from oracle import nacf_oracle as O
from util import maxerr

pytestmark = pytest.mark.gpu


def build(opt, sd, dev, **extra):
    import nacf_amd
    o = dict(opt)
    o.update(extra)
    m = nacf_amd.get_model(o)
    m.load_state_dict({k: v.clone() for k, v in sd.items()})
    return m.to(dev)


class TieRecorder:
    """wraps the oracle's na_step / select_worst: per sequence, was every decision it took clear of round-off?"""

    def __init__(self, n_rows, logit_eps=1e-5, conf_rel=2e-5):
        # logits are O(1): 1e-5 is ~50x the HIP-vs-oracle logit error; confidences are compared RELATIVELY (at random init they
        # are ~1e-4, and neighbouring positions' confidences differ by a few per cent): 2e-5 is ~20x the fp32 error of a
        # soft-max probability
        self.safe = torch.ones(n_rows, dtype=torch.bool)
        self.n_logit = self.n_conf = 0
        self.logit_eps, self.conf_rel = logit_eps, conf_rel
        self._na_step, self._select_worst = O.na_step, O.select_worst

    def __enter__(self):
        rec = self

        def na_step(sd, opt, tokens, enc_output, category, pad_mask):
            h, _, _ = O.decoder_forward(sd, opt, tokens, enc_output, category, decoding_type="NARFormer")
            logits = O.vocab_logits(sd, opt, h)
            top2 = logits.topk(2, dim=-1).values
            margin = (top2[..., 0] - top2[..., 1]).masked_fill(pad_mask, float("inf"))
            ok = margin.min(dim=1).values > rec.logit_eps
            rec.n_logit += int((~ok).sum())
            rec.safe &= ok
            return rec._na_step(sd, opt, tokens, enc_output, category, pad_mask)

        def select_worst(token_probs, num_mask):
            srt = token_probs.sort(dim=1).values
            k = num_mask.clamp(min=1)
            L = srt.shape[1]
            lo = srt.gather(1, (k - 1).clamp(max=L - 1).unsqueeze(1)).squeeze(1)
            hi = srt.gather(1, k.clamp(max=L - 1).unsqueeze(1)).squeeze(1)
            ok = (k >= L) | ((hi - lo) > rec.conf_rel * hi)
            rec.n_conf += int((~ok).sum())
            rec.safe &= ok
            return rec._select_worst(token_probs, num_mask)

        O.na_step, O.select_worst = na_step, select_worst
        return self

    def __exit__(self, *a):
        O.na_step, O.select_worst = self._na_step, self._select_worst
        return False


_ORACLE_CACHE = {}


@pytest.mark.parametrize("graph", ["off", "on"])
@pytest.mark.parametrize("B", [32, 128])
def test_full_size_na_decode_vs_oracle(dev, B, graph):
    import nacf_amd
    from nacf_amd import synthetic as S
    from nacf_amd.models.Translator import Translator
    opt = nacf_amd.opts.make_opt("NACF", "MSRVTT", with_category=True, max_len=20, vocab_size=10547, n_frames=60, fused_loss=True)
    sd = S.init_state_dict(opt, seed=0)
    b = S.synth_batch(opt, B, 60, seed=21)
    model = build(opt, sd, dev)
    model.eval()
    dec = dict(paradigm="mp", use_ct=True, iterations=5, length_beam_size=6, beam_alpha=1.35)
    dopt = dict(model.opt)
    dopt.update(dec)
    dopt.update(collect_best_candidate_iterative_results=True, not_only_best_candidate=True, decode_graph=graph)
    feats, cat = [f.to(dev) for f in b["feats"]], b["category"].to(dev)
    tr = Translator(model, dopt, device=dev)
    with torch.no_grad():
        enc = model.encode(feats=feats)
        hyp, (it_tok, it_prob) = tr.translate_batch(enc, cat, None, None)
        if graph == "on":       # the replayed graph, not only the capture run
            hyp2, (it_tok2, _) = tr.translate_batch(enc, cat, None, None)
            assert torch.equal(hyp, hyp2) and torch.equal(it_tok, it_tok2)
    # the oracle on the host cores, recording where its own decisions were numerically tied (once per batch size: the
    # graph on / off cases share it)
    lbs = dec["length_beam_size"]
    if B not in _ORACLE_CACHE:
        o_enc = O.encode(sd, opt, b["feats"], training=False)
        col = []
        with TieRecorder(B * lbs) as rec:
            o_res = O.generate(sd, opt, dec, o_enc, b["category"], None, col)
        _ORACLE_CACHE[B] = (o_enc, col, rec, o_res)
    o_enc, col, rec, (o_hyp, o_all, o_lp, o_beam) = _ORACLE_CACHE[B]
    assert maxerr(enc["enc_output"], o_enc["enc_output"]) < 2e-4 and maxerr(enc["pred_length"], o_enc["pred_length"]) < 2e-4
    o_tok = torch.stack([c[0] for c in col], 1)
    o_prob = torch.stack([c[1] for c in col], 1)
    assert o_tok.shape == tuple(it_tok.shape), (o_tok.shape, it_tok.shape)
    # the length beam itself (top-k of pred_length) must not be tied either
    pl = o_enc["pred_length"].sort(dim=1, descending=True).values
    beam_safe = (pl[:, lbs - 1] - pl[:, lbs]) > 1e-5
    safe_rows = rec.safe & beam_safe.repeat_interleave(lbs)
    assert float(safe_rows.float().mean()) >= 0.99, (float(safe_rows.float().mean()), rec.n_logit, rec.n_conf, int((~beam_safe).sum()))
    assert torch.equal(it_tok.cpu()[safe_rows], o_tok[safe_rows])
    assert maxerr(it_prob.cpu()[safe_rows], o_prob[safe_rows]) < 1e-4
    # the rows the oracle marks as numerically tied are not bit-exact by construction, but a tie flips one slot of one pass:
    # they must still agree with the oracle almost everywhere (a kernel that scrambled exactly the tied rows would pass the
    # assertions above) -- floor on the share of equal token ids over those rows, all passes
    unsafe_agree = 1.0
    if int((~safe_rows).sum()) > 0:
        unsafe_agree = float((it_tok.cpu()[~safe_rows] == o_tok[~safe_rows]).float().mean())
        assert unsafe_agree >= 0.90, (unsafe_agree, int((~safe_rows).sum()))
    # final choice among the candidates: score = sum log p / len^alpha (na_generate.py:66-77)
    score = o_lp.sum(-1) / (o_beam.float() ** dec["beam_alpha"])
    top2 = score.topk(2, dim=1).values
    safe_vid = safe_rows.view(B, lbs).all(1) & ((top2[:, 0] - top2[:, 1]) > 1e-5)
    # (one video of 32 is 3 %; at 128 videos the ORACLE's own decisions are free of numerical ties for 124 of 128 videos = 0.969 --
    #  a property of the seeded inputs, printed below: a regression of the best-candidate scorer must not hide inside the small
    #  batch's slack)
    assert float(safe_vid.float().mean()) >= (0.96 if B >= 128 else 0.93), float(safe_vid.float().mean())
    assert hyp.shape == o_hyp.shape
    assert torch.equal(hyp.cpu()[safe_vid], o_hyp[safe_vid])
    print("full-size NA decode B=%d graph=%s: %d/%d candidate sequences (%.4f) and %d/%d videos (%.4f) free of numerical ties, all bit-exact; token agreement on the %d tied rows %.4f"
          % (B, graph, int(safe_rows.sum()), safe_rows.numel(), float(safe_rows.float().mean()), int(safe_vid.sum()), B,
             float(safe_vid.float().mean()), int((~safe_rows).sum()), unsafe_agree))


def test_full_size_ar_beam_vs_oracle(dev):
    import nacf_amd
    from nacf_amd import synthetic as S
    from nacf_amd.models.Translator import Translator
    B = 32
    opt = nacf_amd.opts.make_opt("ARB2", "MSRVTT", with_category=True, max_len=20, vocab_size=10547, n_frames=60)
    sd = S.init_state_dict(opt, seed=11)
    sd["tgt_word_prj.weight"][O.EOS] *= 3.0          # random-init models never emit <eos>: make ends happen at mixed steps
    b = S.synth_batch(opt, B, 60, seed=22)
    model = build(opt, sd, dev)
    model.eval()
    dopt = dict(model.opt, beam_size=5, beam_alpha=1.0, topk=1)
    with torch.no_grad():
        enc = model.encode(feats=[f.to(dev) for f in b["feats"]])
        hyps, scores = Translator(model, dopt, device=dev).translate_batch(enc, b["category"].to(dev), None, None)
    o_enc = O.encode(sd, opt, b["feats"], training=False)
    o_h, o_s = O.ar_beam_search(sd, opt, o_enc, b["category"], beam_size=5, alpha=1.0, topk=1)
    assert len(hyps) == B
    same = 0
    lens = set()
    for i in range(B):
        assert len(hyps[i]) == len(o_h[i]) == 1
        if hyps[i][0] == o_h[i][0]:
            same += 1
            assert abs(scores[i][0] - o_s[i][0]) < 5e-4, (i, scores[i][0], o_s[i][0])
        else:       # a near-tie inside the flat top-k over beam x vocabulary may reorder beams: scores must still agree
            assert abs(scores[i][0] - o_s[i][0]) < 2e-3, (i, hyps[i][0], o_h[i][0], scores[i][0], o_s[i][0])
        lens.add(len(o_h[i][0]))
    assert same >= B - 1, same
    differing = [i for i in range(B) if hyps[i][0] != o_h[i][0]]
    print("full-size AR beam-5, B=%d: %d/%d hypotheses identical, lengths seen %s" % (B, same, B, sorted(lens)))
    for i in differing:      # (which one, and why it is tolerated: same score to 2e-3 = two beams the flat top-k saw as a near-tie)
        print("  video %d differs: HIP %s (score %.6f) vs oracle %s (score %.6f)" % (i, hyps[i][0], scores[i][0], o_h[i][0], o_s[i][0]))


def test_ar_beam_at_the_bench_batch_is_the_small_batch_result(dev):
    """BASELINE configs[4] decodes B = 256 videos per batch (bench.py: config5_ar_vs_na); the oracle comparison above runs at
    B = 32 (a CPU beam search of 256 videos takes minutes).  Beam search treats videos independently (models/Beam.py:5-169), so
    the first 32 videos of a 256-video batch must come out exactly as in the 32-video batch: same hypotheses, same scores --
    which extends the oracle parity of the B = 32 test to the bench's batch size."""
    import nacf_amd
    from nacf_amd import synthetic as S
    from nacf_amd.models.Translator import Translator
    opt = nacf_amd.opts.make_opt("ARB2", "MSRVTT", with_category=True, max_len=20, vocab_size=10547, n_frames=60)
    sd = S.init_state_dict(opt, seed=11)
    sd["tgt_word_prj.weight"][O.EOS] *= 3.0
    b = S.synth_batch(opt, 256, 60, seed=22)
    model = build(opt, sd, dev)
    model.eval()
    dopt = dict(model.opt, beam_size=5, beam_alpha=1.0, topk=1)
    out = {}
    for n in (256, 32):
        with torch.no_grad():
            enc = model.encode(feats=[f[:n].to(dev) for f in b["feats"]])
            out[n] = Translator(model, dopt, device=dev).translate_batch(enc, b["category"][:n].to(dev), None, None)
    (h256, s256), (h32, s32) = out[256], out[32]
    assert len(h256) == 256 and len(h32) == 32
    same = sum(1 for i in range(32) if h256[i][0] == h32[i][0])
    # (the GEMM tile choice depends on the row count, so logits differ by summation order: near-ties may flip a beam)
    assert same >= 31, same
    for i in range(32):
        assert abs(s256[i][0] - s32[i][0]) < (5e-4 if h256[i][0] == h32[i][0] else 2e-3), (i, s256[i][0], s32[i][0])
    assert len({len(h[0]) for h in h256}) > 3          # ends at mixed steps


@pytest.mark.parametrize("graph", ["off", "on"])
def test_na_decode_at_the_bench_batch_is_the_small_batch_result(dev, graph):
    """BASELINE configs[4], the NA side: bench.py decodes 256 videos per batch; the oracle comparison above runs at 32 and 128.
    Mask-predict treats videos independently (decoding/na_generate.py:19-131), so the first 128 videos of a 256-video batch must
    come out as in the 128-video batch -- which extends the oracle parity of the B = 128 test to the bench's batch size.  (The GEMM
    tile choice depends on the row count, so logits differ by summation order: a numerically tied argmax may flip a token; the
    share of identical captions is asserted and printed, the per-iteration tokens agree on >= 99.5 % of the slots.)"""
    import nacf_amd
    from nacf_amd import synthetic as S
    from nacf_amd.models.Translator import Translator
    opt = nacf_amd.opts.make_opt("NACF", "MSRVTT", with_category=True, max_len=20, vocab_size=10547, n_frames=60, fused_loss=True)
    sd = S.init_state_dict(opt, seed=0)
    b = S.synth_batch(opt, 256, 60, seed=21)
    model = build(opt, sd, dev)
    model.eval()
    dopt = dict(model.opt)
    dopt.update(paradigm="mp", use_ct=True, iterations=5, length_beam_size=6, beam_alpha=1.35,
                collect_best_candidate_iterative_results=True, not_only_best_candidate=True, decode_graph=graph)
    out = {}
    for n in (256, 128):
        with torch.no_grad():
            enc = model.encode(feats=[f[:n].to(dev) for f in b["feats"]])
            tr = Translator(model, dopt, device=dev)
            hyp, (it_tok, _) = tr.translate_batch(enc, b["category"][:n].to(dev), None, None)
            if graph == "on":
                hyp, (it_tok, _) = tr.translate_batch(enc, b["category"][:n].to(dev), None, None)      # the replayed graph
        out[n] = (hyp.cpu(), it_tok.cpu())
    (h256, t256), (h128, t128) = out[256], out[128]
    assert h256.shape[0] == 256 and h128.shape[0] == 128
    lbs = 6
    same_vid = float((h256[:128] == h128).all(1).float().mean())
    same_tok = float((t256[:128 * lbs] == t128).float().mean())
    print("NA decode, graph %s: first 128 of 256 videos vs the 128-video batch: %.4f of the captions identical, %.5f of the per-iteration tokens"
          % (graph, same_vid, same_tok))
    assert same_vid >= 0.97 and same_tok >= 0.995, (same_vid, same_tok)


def test_config0_nab_youtube2text_shape_train_step_vs_oracle(dev):
    """BASELINE configs[0]: NAB on Youtube2Text-shape features (60 x 2048 image + motion), batch 16"""
    import nacf_amd
    from nacf_amd import synthetic as S
    from nacf_amd.misc.crit import get_criterion
    from nacf_amd.misc.optim import get_optimizer
    B = 16
    opt = nacf_amd.opts.make_opt("NAB", "Youtube2Text", with_category=False, max_len=20, vocab_size=10547, n_frames=60,
                                 fused_loss=True, hidden_dropout_prob=0.0, encoder_dropout=0.0, beta=[0, 1])
    sd = S.init_state_dict(opt, seed=0)
    assert not any("category" in k for k in sd)
    b = S.synth_batch(opt, B, 60, seed=12)
    model = build(opt, sd, dev)
    model.train()
    crit, optim = get_criterion(model.opt), get_optimizer(model.opt, model)
    optim.zero_grad()
    res = model(feats=[f.to(dev) for f in b["feats"]], tgt_tokens=b["tokens"].to(dev), category=b["category"].to(dev))      # ignored: no category table
    res["tgt_word_labels"] = b["labels"].to(dev)
    res["tgt_length"] = b["tgt_length"].to(dev)
    loss = crit.get_loss(res)
    loss.backward()
    grads = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()}

    def oracle(dt):
        s_ = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        l_, _, g_ = O.train_step(s_, opt, [f.to(dt) for f in b["feats"]], b["tokens"], b["category"], b["labels"],
                                 b["tgt_length"].to(dt), {}, lr=opt["learning_rate"])
        return float(l_), g_
    o_loss, o_grads = oracle(torch.float32)
    d_loss, d_grads = oracle(torch.float64)
    assert abs(float(loss) - o_loss) <= 1e-5 * abs(o_loss), (float(loss), o_loss)
    assert abs(float(loss) - d_loss) <= 1e-5 * abs(d_loss), (float(loss), d_loss)
    worst = (0.0, "")
    for k, g in grads.items():
        ref64 = d_grads[k]
        scale = float(ref64.abs().max())
        if scale <= 1e-7:
            assert float(g.abs().max()) < 1e-5, k
            continue
        e64 = float((g.double() - ref64).abs().max()) / scale
        cpu64 = float((o_grads[k].double() - ref64).abs().max()) / scale
        assert e64 <= max(1e-4, 3.0 * cpu64), (k, e64, cpu64)
        worst = max(worst, (e64, k))
    print("configs[0] NAB/Youtube2Text-shape B=16: loss %.6f, worst gradient error vs the double oracle %.2e of max (%s)"
          % (float(loss), worst[0], worst[1]))


def test_ar_beam_last_slot_subset_equals_the_full_prefix_pass(dev):
    """models/Beam.py reads only the last position's hidden state of every step (Translator.py:111), so the last layer's
    query-side work runs on that slot alone (opt['ar_last_slot_only'], default on).  Against the full-prefix pass: the same
    hypotheses, scores to fp32 round-off (the GEMMs of the two forms see different row counts, i.e. tile choices)."""
    import nacf_amd
    from nacf_amd import synthetic as S
    from nacf_amd.models.Translator import Translator
    opt = nacf_amd.opts.make_opt("ARB2", "MSRVTT", with_category=True, max_len=20, vocab_size=10547, n_frames=60)
    sd = S.init_state_dict(opt, seed=11)
    sd["tgt_word_prj.weight"][O.EOS] *= 3.0
    b = S.synth_batch(opt, 32, 60, seed=22)
    model = build(opt, sd, dev)
    model.eval()
    out = {}
    for sub in (True, False):
        dopt = dict(model.opt, beam_size=5, beam_alpha=1.0, topk=1, ar_last_slot_only=sub)
        with torch.no_grad():
            enc = model.encode(feats=[f.to(dev) for f in b["feats"]])
            out[sub] = Translator(model, dopt, device=dev).translate_batch(enc, b["category"].to(dev), None, None)
    (h1, s1), (h0, s0) = out[True], out[False]
    same = sum(1 for i in range(32) if h1[i][0] == h0[i][0])
    assert same >= 31, same
    for i in range(32):
        assert abs(s1[i][0] - s0[i][0]) < (5e-4 if h1[i][0] == h0[i][0] else 2e-3), (i, s1[i][0], s0[i][0])
