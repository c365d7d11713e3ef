"""CPU: the oracle restatement vs the golden vectors captured from the
reference (oracle/make_golden.py).  This is the pin of the oracle."""
import numpy as np
import pytest
import torch

from oracle import nacf_oracle as O
from util import gold_batch, gold_json, gold_opt, gold_state, load_gold, maxerr, t


def _tok_labels(opt, b):
    vw = opt["visual_word_generation"]
    tokens = [b["tokens_1"], b["tokens"]] if vw else b["tokens"]
    labels = [b["labels_1"], b["labels"]] if vw else b["labels"]
    return tokens, labels


@pytest.mark.parametrize("name", ["tiny_nacf_train", "tiny_nab_train", "tiny_arb2_train", "tiny_arb_train", "tiny_arb_watch_train", "tiny_nab_nogate_train", "tiny_nacf_pmlm_train", "tiny_nab_pmlm_ln_train", "tiny_nacf_lwe_train",
                                  "tiny_nab_variants_train", "tiny_nacf_ln_train", "tiny_nacf_pos_train", "tiny_nacf_pos_ln_train"])
def test_train_step_matches_reference(name):
    g = load_gold(name)
    opt = gold_opt(g)
    b = gold_batch(g)
    sd = O.init_state_dict(opt, seed=0)
    tokens, labels = _tok_labels(opt, b)
    # forward outputs
    res = O.forward_train(sd, opt, b["feats"], tokens, b["category"], training=True, new_stats={})
    assert maxerr(res["enc_output"], t(g["out.enc_output"])) < 2e-6
    for i, lp in enumerate(res["tgt_word_logprobs"]):
        assert maxerr(lp, t(g[f"out.logprobs{i}"])) < 5e-6
    if "out.pred_length" in g.files:
        assert maxerr(res["pred_length"], t(g["out.pred_length"])) < 2e-6
    # loss, gradients, one optimiser step
    st = {}
    loss, info, grads = O.train_step(sd, opt, b["feats"], tokens, b["category"], labels, b.get("tgt_length"), st,
                                     lr=opt["learning_rate"])
    assert abs(float(loss) - float(g["loss"])) < 2e-5
    for k, v in gold_state(g, "grad.").items():
        assert maxerr(grads[k].clamp(-opt["grad_clip"], opt["grad_clip"]), v) < 5e-6, k
    for k, v in gold_state(g, "after.").items():
        if v.is_floating_point():
            # Adam's first step is lr*g/(|g|+eps): ill-conditioned where |g| ~ eps, hence 1e-4
            assert maxerr(sd[k], v) < 1e-4, k
        else:
            assert int(sd[k]) == int(v)


@pytest.mark.parametrize("name,seed", [("tiny_nacf_decode", 3), ("tiny_nab_decode", 3), ("tiny_nacf_goldlen_decode", 3)])
def test_decode_matches_reference(name, seed):
    g = load_gold(name)
    opt = gold_opt(g)
    b = gold_batch(g)
    sd = O.init_state_dict(opt, seed=seed)
    enc = O.encode(sd, opt, b["feats"], training=False)
    assert maxerr(enc["enc_output"], t(g["out.enc_output"])) < 2e-6
    assert maxerr(enc["pred_length"], t(g["out.pred_length"])) < 2e-6
    variants = sorted({k.split(".")[0] for k in g.files if k.endswith(".hyp")})
    assert variants
    for v in variants:
        dec = gold_json(g, v + ".dec_json")
        col = []
        hyp, _, lp, beam = O.generate(sd, opt, dec, enc, b["category"], None, col, gold_tokens=b.get("gold_tokens"))
        assert torch.equal(hyp, t(g[v + ".hyp"])), v
        toks = torch.stack([c[0] for c in col], 1)
        assert torch.equal(toks, t(g[v + ".iter_tokens"]).long()), v
        assert maxerr(torch.stack([c[1] for c in col], 1), t(g[v + ".iter_probs"])) < 5e-6


def test_decode_with_teacher_matches_reference():
    g = load_gold("tiny_nacf_teacher")
    opt, t_opt = gold_opt(g), gold_opt(g, "teacher_opt_json")
    b = gold_batch(g)
    sd, t_sd = O.init_state_dict(opt, seed=3), O.init_state_dict(t_opt, seed=7)
    enc = O.encode(sd, opt, b["feats"], training=False)
    t_enc = O.encode(t_sd, t_opt, b["feats"], training=False)
    for v in ("mp_ct", "mp_md"):
        dec = gold_json(g, v + ".dec_json")
        hyp, *_ = O.generate(sd, opt, dec, enc, b["category"], (t_sd, t_opt, t_enc["enc_output"]), None)
        assert torch.equal(hyp, t(g[v + ".hyp"])), v


@pytest.mark.parametrize("name", ["tiny_arb2_beam", "tiny_arb_beam", "tiny_arb_beam_eos", "tiny_arb2_beam_eos", "tiny_arb_watch_beam"])
def test_ar_beam_matches_reference(name):
    g = load_gold(name)
    opt = gold_opt(g)
    b = gold_batch(g)
    sd = O.init_state_dict(opt, seed=11)
    sd["tgt_word_prj.weight"][O.EOS] *= float(g["eos_boost"])
    enc = O.encode(sd, opt, b["feats"], training=False)
    hyps, scores = O.ar_beam_search(sd, opt, enc, b["category"], beam_size=int(g["beam_size"]),
                                    alpha=float(g["alpha"]), topk=int(g["topk"]))
    for i, h in enumerate(hyps):
        assert len(h) == int(g["n_best"][i])
        for j, x in enumerate(h):
            n = int(g["hyp_len"][i, j])
            assert x == g["hyp"][i, j, :n].tolist()
            assert abs(scores[i][j] - float(g["score"][i, j])) < 1e-5


def test_weight_generator_is_reproducible():
    g = load_gold("full_nacf")
    opt = gold_opt(g)
    sd = O.init_state_dict(opt, seed=int(g["seed_weights"]))
    names = [str(n) for n in g["weight_names"]]
    assert names == list(sd.keys())
    sums = np.array([float(v.double().sum()) for v in sd.values()])
    np.testing.assert_allclose(sums, g["weight_checksums"], rtol=0, atol=1e-9)
