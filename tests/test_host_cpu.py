"""CPU (no GPU needed): the C-ABI library loads and exports every symbol the
header declares, host-side logic (option overlays, flat parameter layout,
registry, state_dict contract) behaves like the reference's."""
import os
import re

import pytest
import torch

import nacf_amd
from nacf_amd.runtime import lib
from oracle import nacf_oracle as O
from util import gold_opt, gold_state, load_gold

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol():
    hdr = open(os.path.join(ROOT, "include", "nacf_hip.h")).read()
    declared = set(re.findall(r"^(?:int|size_t|const char\*)\s+(nacf_[a-z0-9_]+)\s*\(", hdr, re.M))
    assert len(declared) >= 45
    handle = lib.load()
    for name in declared:
        assert hasattr(handle, name), name
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)
    assert handle.nacf_version() >= 1
    assert handle.nacf_last_error() is not None
    # the loader self-check: header macro == library's answer == number of declarations == ctypes table
    macro = int(re.search(r"#define\s+NACF_ABI_COUNT\s+(\d+)", hdr).group(1))
    assert macro == len(declared) == handle.nacf_abi_count() == len(lib.SIGNATURES)


def test_no_cpu_fallback():
    opt = gold_opt(load_gold("tiny_nacf_train"))
    m = nacf_amd.get_model(opt)
    with pytest.raises(Exception):
        m(feats=[torch.rand(2, 6, 32), torch.rand(2, 6, 32)], tgt_tokens=[torch.ones(2, 10, dtype=torch.long)] * 2,
          category=torch.zeros(2, 1, dtype=torch.long))


def test_method_overlays_match_reference_flags():
    o = nacf_amd.opts.make_opt("NACF", "MSRVTT", default=True, vocab_size=10)
    assert (o["encoder"], o["decoder"], o["decoding_type"]) == ("Encoder_HighWay", "BertDecoderDisentangled", "NARFormer")
    assert o["visual_word_generation"] and o["use_ct"] and o["max_len"] == 30 and o["with_category"]
    assert o["crit"] == ["lang", "length"] and o["beam_alpha"] == 1.35 and o["length_beam_size"] == 6
    o = nacf_amd.opts.make_opt("ARB", "Youtube2Text", default=True)
    assert o["decoding_type"] == "ARFormer" and o["beam_size"] == 5 and o["max_len"] == 20 and not o["with_category"]
    assert o["crit"] == ["lang"]
    with pytest.raises(AssertionError):
        nacf_amd.opts.make_opt("NOPE")
    with pytest.raises(ValueError):
        nacf_amd.get_model(dict(o, vocab_size=10, decoder="Nope"))


@pytest.mark.parametrize("name", ["tiny_nacf_train", "tiny_nab_train", "tiny_arb2_train", "tiny_arb_train", "tiny_arb_watch_train", "tiny_nab_nogate_train", "tiny_nacf_pmlm_train", "tiny_nab_pmlm_ln_train", "tiny_nacf_lwe_train",
                                  "tiny_nab_variants_train", "tiny_nacf_ln_train", "tiny_nacf_pos_train", "tiny_nacf_pos_ln_train"])
def test_state_dict_contract_and_flat_layout(name):
    g = load_gold(name)
    opt = gold_opt(g)
    m = nacf_amd.get_model(opt)
    ref = gold_state(g, "after.")
    sd = m.state_dict()
    assert set(sd) == set(ref)
    for k, v in ref.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    m.load_state_dict(ref)
    # every parameter is a view into the single flat buffer, gradients into the second one
    lo, hi = m.flat.data.data_ptr(), m.flat.data.data_ptr() + m.flat.total * 4
    for k, p in m.named_parameters():
        assert lo <= p.data_ptr() < hi, k
        assert p.grad is not None and m.flat.grad.data_ptr() <= p.grad.data_ptr() < m.flat.grad.data_ptr() + m.flat.total * 4
        assert torch.equal(p.detach(), ref[k])
    # packed operands are plain slices
    dec = m.decoder.bert if hasattr(m.decoder, "bert") else m.decoder
    layer = dec.layer[0]
    D = opt["dim_hidden"]
    qkv = layer._pk["qkv"].w
    assert torch.equal(qkv[:D], layer.attention.self.query.weight) and torch.equal(qkv[2 * D:], layer.attention.self.value.weight)
    ckv = layer._pk["ckv"]
    assert torch.equal(ckv.w[D:], layer.attend_to_enc_output.self.value.weight)
    assert torch.equal(ckv.b[:D], layer.attend_to_enc_output.self.key.bias)
    if opt.get("gate", True):
        hw = m.encoder._cfg[0]["hw"].w
        assert torch.equal(hw[D:], m.encoder.Encoder_M[1].w2.weight)
    else:                               # HighWay without its gate: w1 alone (models/Encoder.py:13-15)
        assert not hasattr(m.encoder.Encoder_M[1], "w2")
        assert torch.equal(m.encoder._cfg[0]["hw"]["pack"].w, m.encoder.Encoder_M[1].w1.weight)
    # zero_grad / re-attach semantics
    for p in m.parameters():
        p.grad = None
    assert not m.flat.grads_attached()
    m._ensure_grads()
    assert m.flat.grads_attached() and float(m.flat.grad.abs().sum()) == 0.0


def test_default_init_consumes_rng_like_the_reference_layout():
    # same seed -> same weights twice, and the PAD row of the word embedding is zero (padding_idx)
    opt = gold_opt(load_gold("tiny_nacf_train"))
    torch.manual_seed(0)
    a = nacf_amd.get_model(opt).state_dict()
    torch.manual_seed(0)
    b = nacf_amd.get_model(opt).state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert float(a["decoder.bert.embedding.word_embeddings.weight"][0].abs().sum()) == 0.0
    assert set(a) == set(O.param_shapes(opt))


def test_unsupported_variants_fail_loudly():
    opt = gold_opt(load_gold("tiny_nacf_train"))
    for bad in (dict(enhance_input=1), dict(fusion="addition"), dict(hidden_act="swish")):
        with pytest.raises((NotImplementedError, ValueError)):
            nacf_amd.get_model(dict(opt, **bad))


def test_num_mask_lut_matches_torch_float_semantics():
    from nacf_amd.decoding.algorithms import Algorithm_Base
    alg = Algorithm_Base({}, {}, None)
    T, Lp = 6, 29
    lut = alg.num_mask_lut([1.0 - (c / T) for c in range(T)], Lp, "cpu")
    for c in range(T):
        lens = torch.arange(Lp + 1)
        assert torch.equal(lut[c].long(), (lens.float() * (1.0 - (c / T))).long())


def test_reference_checkpoint_contract(tmp_path):
    """a checkpoint file WRITTEN BY THE REFERENCE (tests/golden/tiny_arb_checkpoint.pth.tar, made by
    oracle/make_golden.py with the reference's save_checkpoint) loads through the mirrored helpers, and the teacher-init
    remap takes exactly the tensors the reference's load_satisfied_weights took (misc/utils.py:54-63,158-192)."""
    import os
    import numpy as np
    from nacf_amd.misc.utils import load_model_and_opt, load_satisfied_weights, save_checkpoint
    from util import GOLD
    g = np.load(os.path.join(GOLD, "tiny_checkpoint.npz"))
    path = os.path.join(GOLD, "tiny_arb_checkpoint.pth.tar")
    model, opt, other = load_model_and_opt(path, "cpu", return_other_info=True)
    assert sorted(other) == ["epoch", "settings", "validate_result"] and other["epoch"] == int(g["epoch"])
    assert opt["method"] == "ARB" and type(model.decoder).__name__ == "BertDecoder"
    raw = torch.load(path, map_location="cpu", weights_only=False)["state_dict"]
    sd = model.state_dict()
    assert set(sd) == set(raw) and all(torch.equal(sd[k], raw[k]) for k in raw)
    # teacher init of a NACF student: 'decoder.bert.X' <- 'decoder.X'
    import json
    s_opt = json.loads(str(g["student_opt_json"]))
    student = nacf_amd.get_model(s_opt)
    from nacf_amd import synthetic as S
    student.load_state_dict(S.init_state_dict(s_opt, seed=6))
    before = {k: v.clone() for k, v in student.state_dict().items()}
    load_satisfied_weights(student, path, str_mapping={"decoder.bert.": "decoder."})
    after = student.state_dict()
    names = [str(n) for n in g["names"]]
    assert names == list(after.keys())
    taken = [not torch.equal(after[k], before[k]) for k in names]
    assert taken == [bool(x) for x in g["taken"]] and sum(taken) == int(g["n_taken"])
    np.testing.assert_allclose([float(after[k].double().sum()) for k in names], g["sums"], rtol=0, atol=1e-9)
    for k in names:
        if k.startswith("decoder.bert.") and k.replace("decoder.bert.", "decoder.") in raw:
            assert torch.equal(after[k], raw[k.replace("decoder.bert.", "decoder.")]), k
    with pytest.raises(AssertionError):
        load_satisfied_weights(nacf_amd.get_model(s_opt), path, strict=True)          # no remap: decoder.bert.* missing
    # round trip through save_checkpoint (same dict layout as misc/run.py:334-339)
    save_checkpoint({"epoch": 1, "state_dict": student.state_dict(), "validate_result": {}, "settings": s_opt}, True,
                    filepath=str(tmp_path), filename="c.pth.tar")
    m2, o2 = load_model_and_opt(os.path.join(str(tmp_path), "best.pth.tar"), "cpu")
    assert all(torch.equal(v, after[k]) for k, v in m2.state_dict().items())
