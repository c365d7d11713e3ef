"""Generate tests/golden/*.npz by IMPORTING the reference (build container only).

TEST INFRASTRUCTURE.  Run from anywhere:  ``python oracle/make_golden.py``.
Needs /root/reference (read-only, never shipped).  What it does:

  1. builds the reference model through its own factory
     (models.get_model(opt), models/__init__.py:64-94) for tiny and full-shape
     configs, loads weights produced by ``nacf_oracle.init_state_dict`` (seeded,
     regenerable anywhere), and runs the reference's forward / criterion /
     backward / optimiser step / Translator.translate_batch on seeded inputs;
  2. asserts the oracle restatement agrees with the reference (<=2e-6 on
     floats, exact on token ids) -- this is the pin;
  3. stores inputs + reference outputs as plain arrays (no pickled objects)
     under tests/golden/.

The fixtures are data (inputs, expected outputs); no reference source text is
stored.  SURVEY.md section 8c lists the cases.
"""
import copy
import io
import json
import os
import sys
import contextlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
torch.set_num_threads(8)

from oracle import nacf_oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)


def ref_opt(method, dataset="MSRVTT", extra=()):
    """Reference option dict via its own opts.parse_opt (opts.py:5-213)."""
    cwd = os.getcwd()
    os.chdir(REF)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import opts  # noqa
    argv = sys.argv
    sys.argv = ["train.py", "--method", method, "--dataset", dataset, "--scope", "x"] + list(extra)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            opt = vars(opts.parse_opt())
    finally:
        sys.argv = argv
        os.chdir(cwd)
    return opt


def ref_model(opt, sd):
    os.chdir(REF)
    import models  # noqa
    with contextlib.redirect_stdout(io.StringIO()):
        m = models.get_model(opt)
    os.chdir(ROOT)
    missing = set(m.state_dict().keys()) ^ set(sd.keys())
    assert not missing, missing
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), (k, v.shape, sd[k].shape)
    m.load_state_dict({k: v.clone() for k, v in sd.items()})
    return m


TINY = ["--dim_hidden", "64", "--num_attention_heads", "4", "--intermediate_size", "128",
        "--dim_i", "32", "--dim_m", "32", "--max_len", "10",
        "--hidden_dropout_prob", "0", "--encoder_dropout", "0", "--n_frames", "6"]


def keep(opt):
    ks = list(O.DEFAULT_OPT.keys()) + ["vocab_size", "grad_clip", "weight_decay", "learning_rate", "crit", "minimum_learning_rate", "decay", "optim"]
    return {k: opt[k] for k in ks if k in opt}


def check(a, b, tol, what):
    err = float((a - b).abs().max())
    assert err <= tol, f"{what}: oracle vs reference {err:.3e} > {tol}"
    return err


def npz(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.detach().cpu().numpy()
        else:
            out[k] = np.asarray(v)
    return out


def opt_blob(opt):
    import json
    return np.frombuffer(json.dumps(keep(opt), sort_keys=True).encode(), dtype=np.uint8)


def train_case(name, method, extra, V, B, F_, dataset="MSRVTT", beta=(0.35, 0.9), override=None):
    """forward + loss + backward + clip + Adam, reference vs oracle (dropout 0).  override: option keys the reference reads
    from its opt dictionary but has no command-line switch for (e.g. opt['gate'], models/Encoder.py:64)."""
    opt = ref_opt(method, dataset, TINY + list(extra))
    opt["vocab_size"] = V
    opt.update(override or {})
    sd = O.init_state_dict(opt, seed=0)
    model = ref_model(opt, sd)
    model.train()
    batch = O.synth_batch(opt, B, F_, seed=1, beta=beta)
    os.chdir(REF)
    from misc.crit import get_criterion
    from misc.optim import get_optimizer
    os.chdir(ROOT)
    crit = get_criterion(opt)
    crit.reset_loss_recorder()
    optim = get_optimizer(opt, model)
    vw = opt["visual_word_generation"]
    tokens = [batch["tokens_1"], batch["tokens"]] if vw else batch["tokens"]
    start = 0 if opt["decoding_type"] == "NARFormer" else 1
    labels = [batch["labels_1"], batch["labels"]] if vw else batch["labels"]
    # --- reference step (misc/run.py:254-261) ---
    optim.zero_grad()
    results = model(feats=[f.clone() for f in batch["feats"]], tgt_tokens=tokens, category=batch["category"])
    if opt["decoding_type"] == "NARFormer":
        results["tgt_length"] = batch["tgt_length"]
    results["tgt_word_labels"] = labels
    loss = crit.get_loss(results)
    loss.backward()
    torch.nn.utils.clip_grad_value_(model.parameters(), opt["grad_clip"])
    ref_grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    optim.step()
    ref_after = {k: v.clone() for k, v in model.state_dict().items()}
    names, infos = crit.get_loss_info()
    # --- oracle step ---
    sd_o = {k: v.clone() for k, v in sd.items()}
    st = {}
    o_loss, o_info, o_grads = O.train_step(sd_o, opt, batch["feats"], tokens, batch["category"],
                                           labels, batch.get("tgt_length"), st, lr=opt["learning_rate"])
    errs = {"loss": check(o_loss, loss.detach(), 2e-5, "loss")}
    for k in ref_grads:
        g = o_grads[k].clamp(-opt["grad_clip"], opt["grad_clip"])
        errs["g:" + k] = check(g, ref_grads[k], 5e-6, "grad " + k)
    for k in ref_after:
        if ref_after[k].is_floating_point():
            errs["w:" + k] = check(sd_o[k], ref_after[k], 1e-4, "post-step " + k)  # Adam step 1 = lr*g/(|g|+eps): ill-conditioned where |g|~eps
        else:
            assert int(sd_o[k]) == int(ref_after[k]), k
    # forward outputs (reference values stored)
    model2 = ref_model(opt, sd); model2.train()
    with torch.no_grad():
        r2 = model2(feats=[f.clone() for f in batch["feats"]], tgt_tokens=tokens, category=batch["category"])
    out = {"opt_json": opt_blob(opt), "B": B, "F": F_,
           "loss": loss.detach(), "loss_names": np.array(names), "loss_info": np.array(infos, dtype=np.float64)}
    for k, v in batch.items():
        if k == "feats":
            for i, f in enumerate(v):
                out[f"in.feats{i}"] = f
        else:
            out["in." + k] = v
    out["out.enc_output"] = r2["enc_output"]
    if "pred_length" in r2:
        out["out.pred_length"] = r2["pred_length"]
    for i, lp in enumerate(r2["tgt_word_logprobs"]):
        out[f"out.logprobs{i}"] = lp
    for k, v in ref_grads.items():
        out["grad." + k] = v
    for k, v in ref_after.items():
        out["after." + k] = v
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **npz(out))
    print(f"[{name}] ok  max err {max(errs.values()):.2e}  loss {float(loss):.6f}")
    return opt, sd


def decode_case(name, method, extra, V, B, F_, variants, teacher_method=None, dataset="MSRVTT", gold=False):
    opt = ref_opt(method, dataset, TINY + list(extra))
    opt["vocab_size"] = V
    sd = O.init_state_dict(opt, seed=3)
    # sharpen the distributions a little so decode is not dominated by ties
    model = ref_model(opt, sd)
    model.eval()
    batch = O.synth_batch(opt, B, F_, seed=5)
    os.chdir(REF)
    from models.Translator import Translator
    os.chdir(ROOT)
    out = {"opt_json": opt_blob(opt), "B": B, "F": F_, "in.category": batch["category"]}
    for i, f in enumerate(batch["feats"]):
        out[f"in.feats{i}"] = f
    if gold:        # the captions whose lengths seed the beam under opt['load_generated_captions']
        out["in.gold_tokens"] = batch["tokens"]
    teacher = None
    t_model = None
    if teacher_method is not None:
        t_opt = ref_opt(teacher_method, dataset, TINY + list(extra))
        t_opt["vocab_size"] = V
        t_sd = O.init_state_dict(t_opt, seed=7)
        t_model = ref_model(t_opt, t_sd); t_model.eval()
        out["teacher_opt_json"] = opt_blob(t_opt)
        with torch.no_grad():
            t_enc = t_model.encode(feats=[f.clone() for f in batch["feats"]])
        teacher = (t_sd, t_opt, t_enc["enc_output"])
    with torch.no_grad():
        enc = model.encode(feats=[f.clone() for f in batch["feats"]])
    o_enc = O.encode(sd, opt, batch["feats"], training=False)
    check(o_enc["enc_output"], enc["enc_output"], 2e-6, "enc_output")
    check(o_enc["pred_length"], enc["pred_length"], 2e-6, "pred_length")
    out["out.enc_output"] = enc["enc_output"]
    out["out.pred_length"] = enc["pred_length"]
    for vname, dec in variants.items():
        dopt = dict(opt)
        dopt.update(dec)
        dopt.update(collect_best_candidate_iterative_results=True, not_only_best_candidate=True)
        tr = Translator(model, dopt, device=torch.device("cpu"), teacher_model=t_model)
        with torch.no_grad():
            hyp, (it_tok, it_prob) = tr.translate_batch(
                {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in enc.items()},
                batch["category"], batch["tokens"].clone() if gold else None, {i: str(i) for i in range(V)},
                teacher_encoder_outputs=(t_enc if t_model is not None else None))
        col = []
        o_hyp, o_all, o_lp, o_beam = O.generate(sd, opt, dec, o_enc, batch["category"], teacher, col,
                                                gold_tokens=batch["tokens"] if gold else None)
        assert torch.equal(o_hyp, hyp), f"{name}/{vname}: hypotheses differ"
        o_tok = torch.stack([c[0] for c in col], 1)
        o_prob = torch.stack([c[1] for c in col], 1)
        assert o_tok.shape == it_tok.shape, (o_tok.shape, it_tok.shape)
        assert torch.equal(o_tok, it_tok), f"{name}/{vname}: per-iteration tokens differ"
        check(o_prob, it_prob, 5e-6, f"{name}/{vname} probs")
        out[f"{vname}.hyp"] = hyp
        out[f"{vname}.iter_tokens"] = it_tok.to(torch.int16)
        out[f"{vname}.iter_probs"] = it_prob
        out[f"{vname}.beam"] = o_beam
        out[f"{vname}.cand_lprobs"] = o_lp
        import json
        out[f"{vname}.dec_json"] = np.frombuffer(json.dumps(dec, sort_keys=True).encode(), dtype=np.uint8)
        print(f"[{name}/{vname}] ok  iters {it_tok.shape[1]}  L' {it_tok.shape[2]}")
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **npz(out))


def ar_case(name, method, extra, V, B, F_, dataset="MSRVTT", beam_size=3, topk=1, alpha=1.0, eos_boost=1.0):
    """AR beam search (models/Beam.py + Translator.translate_batch_ARFormer).  `eos_boost` scales the
    <eos> row of tgt_word_prj so that some hypotheses really end with <eos> at random init."""
    opt = ref_opt(method, dataset, TINY + list(extra))
    opt["vocab_size"] = V
    opt["beam_size"] = beam_size
    opt["beam_alpha"] = alpha
    opt["topk"] = topk
    sd = O.init_state_dict(opt, seed=11)
    sd["tgt_word_prj.weight"][O.EOS] *= eos_boost
    model = ref_model(opt, sd); model.eval()
    batch = O.synth_batch(opt, B, F_, seed=13)
    os.chdir(REF)
    from models.Translator import Translator
    os.chdir(ROOT)
    with torch.no_grad():
        enc = model.encode(feats=[f.clone() for f in batch["feats"]])
    tr = Translator(model, opt, device=torch.device("cpu"))
    hyps, scores = tr.translate_batch({k: v for k, v in enc.items()}, batch["category"], None, None)
    o_enc = O.encode(sd, opt, batch["feats"], training=False)
    o_h, o_s = O.ar_beam_search(sd, opt, o_enc, batch["category"], beam_size=beam_size, alpha=alpha, topk=topk)
    assert hyps == o_h, (hyps, o_h)
    for a, b in zip(scores, o_s):
        assert len(a) == len(b) and all(abs(x - y) < 1e-5 for x, y in zip(a, b))
    nb = max(len(h) for h in hyps)
    Lm = max(len(x) for h in hyps for x in h)
    arr = np.zeros((B, nb, Lm), dtype=np.int64)
    lens = np.zeros((B, nb), dtype=np.int64)
    sc = np.zeros((B, nb), dtype=np.float64)
    for i, h in enumerate(hyps):
        for j, x in enumerate(h):
            arr[i, j, :len(x)] = x; lens[i, j] = len(x); sc[i, j] = scores[i][j]
    out = {"opt_json": opt_blob(opt), "B": B, "F": F_, "in.category": batch["category"],
           "hyp": arr, "hyp_len": lens, "score": sc, "n_best": np.array([len(h) for h in hyps]),
           "beam_size": beam_size, "topk": topk, "alpha": alpha, "eos_boost": eos_boost}
    for i, f in enumerate(batch["feats"]):
        out[f"in.feats{i}"] = f
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **npz(out))
    print(f"[{name}] ok  lens {lens.tolist()}  n_eos {int(sum(x[-1] == O.EOS for h in hyps for x in h))}")


def full_shape_case(name="full_nacf", B=4, V=10547, L=20, F_=60):
    """d=512 / F=60 / V=10547: weights come from the seeded generator (only the
    seed + checksums are stored), outputs are sampled."""
    opt = ref_opt("NACF", "MSRVTT", ["-wc", "--max_len", str(L), "--n_frames", str(F_)])
    opt["vocab_size"] = V
    sd = O.init_state_dict(opt, seed=0)
    model = ref_model(opt, sd); model.eval()
    batch = O.synth_batch(opt, B, F_, seed=1)
    os.chdir(REF)
    from models.Translator import Translator
    os.chdir(ROOT)
    with torch.no_grad():
        enc = model.encode(feats=[f.clone() for f in batch["feats"]])
        hid, *_ = model.decoder(batch["tokens"], enc_output=enc["enc_output"], category=batch["category"])
        logits = model.tgt_word_prj(hid)
    o_enc = O.encode(sd, opt, batch["feats"], training=False)
    check(o_enc["enc_output"], enc["enc_output"], 5e-6, "full enc_output")
    o_h, _, _ = O.decoder_forward(sd, opt, batch["tokens"], o_enc["enc_output"], batch["category"])
    check(O.vocab_logits(sd, opt, o_h), logits, 2e-5, "full logits")
    dec = dict(paradigm="mp", use_ct=True, iterations=5, length_beam_size=6, beam_alpha=1.35)
    dopt = dict(opt); dopt.update(dec)
    dopt.update(collect_best_candidate_iterative_results=True, not_only_best_candidate=True)
    tr = Translator(model, dopt, device=torch.device("cpu"))
    with torch.no_grad():
        hyp, (it_tok, it_prob) = tr.translate_batch({k: v.clone() for k, v in enc.items()}, batch["category"],
                                                    None, {i: str(i) for i in range(V)})
    col = []
    o_hyp, _, o_lp, o_beam = O.generate(sd, opt, dec, o_enc, batch["category"], None, col)
    assert torch.equal(o_hyp, hyp)
    assert torch.equal(torch.stack([c[0] for c in col], 1), it_tok)
    g = torch.Generator().manual_seed(99)
    flat = logits.reshape(-1)
    idx = torch.randint(0, flat.numel(), (4096,), generator=g)
    top2 = logits.topk(2, dim=-1)[0]
    out = {"opt_json": opt_blob(opt), "B": B, "F": F_, "seed_weights": 0, "seed_batch": 1,
           "weight_checksums": np.array([float(v.double().sum()) for v in sd.values()]),
           "weight_names": np.array(list(sd.keys())),
           "logit_idx": idx, "logit_val": flat[idx], "logit_margin": (top2[..., 0] - top2[..., 1]),
           "logit_argmax": logits.argmax(-1).to(torch.int16),
           "enc_output_sample": enc["enc_output"][:, ::7, ::5].contiguous(),
           "pred_length": enc["pred_length"],
           "mp_ct.hyp": hyp, "mp_ct.iter_tokens": it_tok.to(torch.int16), "mp_ct.iter_probs": it_prob,
           "mp_ct.beam": o_beam, "in.tokens": batch["tokens"], "in.category": batch["category"]}
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **npz(out))
    print(f"[{name}] ok  min top1-top2 margin {float(out['logit_margin'].min()):.3e}")


EOS_BOOST = float(os.environ.get("EOS_BOOST", "8.0"))


def checkpoint_case():
    """a checkpoint written by the REFERENCE (misc/utils.py:195-202 layout, run.py:334-339 keys)
    plus what the reference's own loaders make of it: load_model_and_opt (utils.py:54-63) and the teacher-init remap
    load_satisfied_weights(..., {'decoder.bert.': 'decoder.'}) (run.py:275-283) into a NACF student."""
    opt_t = ref_opt("ARB", "MSRVTT", TINY + ["-wc"]); opt_t["vocab_size"] = 101
    teacher = ref_model(opt_t, O.init_state_dict(opt_t, seed=5))
    os.chdir(REF)
    from misc.utils import load_model_and_opt, load_satisfied_weights, save_checkpoint
    os.chdir(ROOT)
    path = os.path.join(GOLD, "tiny_arb_checkpoint.pth.tar")
    save_checkpoint({"epoch": 7, "state_dict": teacher.state_dict(), "validate_result": {"CIDEr": 0.5, "loss": 1.25},
                     "settings": opt_t}, False, filepath=GOLD, filename="tiny_arb_checkpoint.pth.tar")
    model, opt_l, other = load_model_and_opt(path, "cpu", return_other_info=True)
    assert sorted(other) == ["epoch", "settings", "validate_result"]
    batch = O.synth_batch(opt_t, 3, 6, seed=2)
    model.eval()
    with torch.no_grad():
        enc = model.encoder(batch["feats"])
    opt_s = ref_opt("NACF", "MSRVTT", TINY + ["-wc"]); opt_s["vocab_size"] = 101
    student = ref_model(opt_s, O.init_state_dict(opt_s, seed=6))
    before = {k: v.clone() for k, v in student.state_dict().items()}
    student = load_satisfied_weights(student, path, str_mapping={"decoder.bert.": "decoder."})
    after = student.state_dict()
    names = list(after.keys())
    taken = np.array([not torch.equal(after[k], before[k]) for k in names])
    np.savez_compressed(os.path.join(GOLD, "tiny_checkpoint.npz"),
                        teacher_opt_json=json.dumps(opt_t), student_opt_json=json.dumps(opt_s),
                        names=np.array(names), taken=taken,
                        sums=np.array([float(after[k].double().sum()) for k in names]),
                        epoch=7, n_taken=int(taken.sum()))
    print("[checkpoint] ok: %d/%d student tensors taken from the ARB checkpoint" % (int(taken.sum()), len(names)))


def _reference_dataset_class():
    """The reference's VideoDataset with its real method bodies.  dataloader.py imports h5py (absent here) at module
    level for the HDF5 reads this fixture does not exercise, so the module source is executed without that one line;
    nothing else is changed and no stand-in is provided."""
    src = open(os.path.join(REF, "dataloader.py")).read().replace("import h5py\n", "")
    ns = {"__name__": "reference_dataloader"}
    cwd = os.getcwd()
    os.chdir(REF)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    try:
        exec(compile(src, os.path.join(REF, "dataloader.py"), "exec"), ns)
    finally:
        os.chdir(cwd)
    return ns


def data_case():
    """the reference's batch construction on a synthetic corpus (dataloader.py:24-37,166-175,
    296-327,329-425), oracle/nacf_data_oracle.py asserted equal, everything saved as arrays."""
    from oracle import nacf_data_oracle as D
    ns = _reference_dataset_class()
    VD = ns["VideoDataset"]
    rs = np.random.RandomState(123)
    tags = ["NOUN", "VERB", "DET", "ADJ", "ADP", "PRON"]
    itop = {i: t for i, t in enumerate(["<pad>", "<unk>", "<bos>", "<eos>", "<mask>", "<vis>"] + tags)}
    V = 101
    itow = {i: "w%d" % i for i in range(V)}
    for i, w in enumerate(D.BE_VERBS):
        itow[6 + i] = w
    demand = ["VERB", "NOUN"]
    demanded = np.array([itop[i] in demand for i in range(len(itop))])
    is_be = np.array([itow[i] in D.BE_VERBS for i in range(V)])
    caps, poss = [], []
    for n in [1, 2, 3, 5, 8, 9, 12, 17, 20, 28, 33]:              # words per caption (max_len 10 / 30 cut some)
        words = rs.randint(6, V, size=n).tolist()
        caps.append([D.BOS] + words + [D.EOS])
        poss.append([2] + rs.randint(6, 6 + len(tags), size=n).tolist() + [3])
    out = {}
    for dt, vw, max_len, beta in [("NARFormer", True, 10, [0.35, 0.9]), ("NARFormer", False, 30, [0.0, 1.0]),
                                  ("ARFormer", True, 10, [0, 1]), ("ARFormer", False, 30, [0, 1])]:
        for mode in ("train", "validate"):
            opt = dict(decoding_type=dt, visual_word_generation=vw, max_len=max_len, beta=beta, demand=demand, seed=7)
            ds = object.__new__(VD)                       # no corpus / HDF5 files: only the pure methods are used
            ds.opt, ds.mode, ds.itow, ds.itop = opt, mode, itow, itop
            ds.random = np.random.RandomState(opt["seed"])
            mine = np.random.RandomState(opt["seed"])
            key = "%s.%s.%d.%s" % (dt, "vw" if vw else "plain", max_len, mode)
            rows = {k: [] for k in ("tokens", "labels", "tokens_1", "labels_1")}
            for c, p in zip(caps, poss):
                if dt == "NARFormer" or True:
                    ref = ds._make_source_target(list(c), list(p))
                got = D.make_source_target(c, p, opt, demanded, is_be, mode == "train", rng=mine)
                pairs = [("tokens", "dec_source"), ("labels", "dec_target")]
                if vw:
                    pairs += [("tokens_1", "dec_source_1"), ("labels_1", "dec_target_1")]
                for mk, rk in pairs:
                    assert list(got[mk]) == list(ref[rk]), (key, mk, got[mk], ref[rk])
                    rows[mk].append(list(ref[rk]))
            for mk, v in rows.items():
                if v and len({len(r) for r in v}) == 1:
                    out[key + "." + mk] = np.array(v, dtype=np.int64)
    # frame sampling + resampling + length targets
    fr = []
    for total, n in [(60, 8), (60, 60), (28, 8), (9, 8), (100, 12), (8, 8)]:
        ref = ns["get_frame_ids"](total, n, "equally_sampling")
        assert ref == D.get_frame_ids(total, n, "equally_sampling")
        fr.append((total, n, ref))
    np.random.seed(11)
    seg_ref = [ns["get_frame_ids"](60, 8, "segment_random") for _ in range(4)]
    mine = np.random.RandomState(11)
    assert seg_ref == [D.get_frame_ids(60, 8, "segment_random", mine) for _ in range(4)]
    rsmp = [(s, t_, ns["resampling"](s, t_)) for s, t_ in [(5, 8), (3, 60), (59, 60), (2, 8)]]
    for s_, t_, r in rsmp:
        assert r == D.resampling(s_, t_)
    lens = [[0, 0, 3, 5, 0, 2], [1] + [0] * 40, [0, 4]]
    lt = []
    for li in lens:
        for max_len in (10, 4):
            ref = list(li)[:max_len]
            ref += [0] * (max_len - len(ref))
            ref = np.array(ref) / sum(ref)                  # dataloader.py:170-175, verbatim arithmetic
            assert np.array_equal(ref, D.length_target(li, max_len))
            lt.append(ref)
    # sample enumeration (_make_infoset, dataloader.py:146-199) on a 6-video corpus
    corpus_caps, corpus_tags, length_info = {}, {}, {}
    for v in range(6):
        k = 3 + v % 3
        corpus_caps["video%d" % v] = [[D.BOS] + rs.randint(6, V, size=rs.randint(2, 9)).tolist() + [D.EOS] for _ in range(k)]
        corpus_tags["video%d" % v] = [[2] + rs.randint(6, 6 + len(tags), size=len(c) - 2).tolist() + [3] for c in corpus_caps["video%d" % v]]
        hist = [0] * 12
        for c in corpus_caps["video%d" % v]:
            hist[len(c) - 2] += 1
        length_info["video%d" % v] = hist
    itoc = {v: (v * 7) % 20 for v in range(6)}
    splits = {"train": [4, 0, 2, 5], "validate": [1], "test": [3]}
    info_rows = {}
    for mode, ncap in [("train", 0), ("train", 2), ("validate", 0)]:
        opt = dict(max_len=10, n_caps_per_video=ncap, seed=3, dataset="x")
        ds = object.__new__(VD)
        ds.opt, ds.mode, ds.specific = opt, mode, -1
        ds.captions, ds.pos_tags, ds.itoc, ds.length_info, ds.splits = corpus_caps, corpus_tags, itoc, length_info, splits
        ds.random = np.random.RandomState(opt["seed"])
        ds.n_caps_per_video = ncap if mode == "train" else 1
        with contextlib.redirect_stdout(io.StringIO()):
            infoset = ds._make_infoset()
        info_rows["%s.%d" % (mode, ncap)] = [[int(it["vid"][5:]), int(it["cap_id"]), int(it["category"])] +
                                            [float(x) for x in it["length_target"]] for it in infoset]
    np.savez_compressed(os.path.join(GOLD, "tiny_data.npz"),
                        corpus_json=json.dumps(dict(captions=corpus_caps, pos_tags=corpus_tags, length_info=length_info,
                                                    itoc={str(k): v for k, v in itoc.items()}, splits=splits,
                                                    itow={str(k): v for k, v in itow.items()},
                                                    itop={str(k): v for k, v in itop.items()})),
                        infoset_json=json.dumps(info_rows),
                        caps=np.array([c + [-1] * (40 - len(c)) for c in caps]), poss=np.array([p + [-1] * (40 - len(p)) for p in poss]),
                        cap_len=np.array([len(c) for c in caps]), demanded=demanded, is_be=is_be,
                        frames_json=json.dumps(fr), seg_random_json=json.dumps(seg_ref), resampling_json=json.dumps(rsmp),
                        length_info_json=json.dumps(lens), **out)
    print("[data] ok:", len(out), "token/label tables,", len(fr), "frame-id cases")


def trajectory_case(name="tiny_nacf_trajectory", method="NACF", extra=("-wc",), V=101, B=4, F_=6, epochs=2, steps=3):
    """the reference's epoch loop (run_train, misc/run.py:249-269, called from train_network_all
    :312-318) for `epochs` x `steps` iterations on seeded batches: zero_grad, forward, get_loss, backward,
    clip_grad_value_, ScheduledOptim.step; the per-epoch loss record and epoch_update_learning_rate between epochs.
    Stored: every batch, the loss of every step, the learning rate of every step, the loss-info of every epoch, the
    final weights, and per parameter the entries whose reference gradient stayed above round-off in every step (Adam
    normalises the others to +-lr steps of arbitrary sign, in the reference too).  The oracle replays the same
    trajectory as the pin."""
    opt = ref_opt(method, "MSRVTT", TINY + list(extra))
    opt["vocab_size"] = V
    sd = O.init_state_dict(opt, seed=0)
    model = ref_model(opt, sd)
    os.chdir(REF)
    from misc.crit import get_criterion
    from misc.optim import get_optimizer
    os.chdir(ROOT)
    crit = get_criterion(opt)
    optim = get_optimizer(opt, model)
    vw = opt["visual_word_generation"]
    sd_o, st_o = {k: v.clone() for k, v in sd.items()}, {}
    out = {"opt_json": opt_blob(opt), "B": B, "F": F_, "epochs": epochs, "steps": steps}
    losses, lrs, infos, solid = [], [], [], None
    seed = 10
    lr_o = opt["learning_rate"]
    for ep in range(epochs):
        model.train()
        crit.reset_loss_recorder()
        for it in range(steps):
            seed += 1
            batch = O.synth_batch(opt, B, F_, seed=seed)
            tokens = [batch["tokens_1"], batch["tokens"]] if vw else batch["tokens"]
            labels = [batch["labels_1"], batch["labels"]] if vw else batch["labels"]
            optim.zero_grad()
            results = model(feats=[f.clone() for f in batch["feats"]], tgt_tokens=tokens, category=batch["category"])
            results["tgt_length"] = batch["tgt_length"]
            results["tgt_word_labels"] = labels
            loss = crit.get_loss(results, epoch=ep)
            loss.backward()
            torch.nn.utils.clip_grad_value_(model.parameters(), opt["grad_clip"])
            big = {k: p.grad.abs() > 2e-6 for k, p in model.named_parameters()}
            solid = big if solid is None else {k: solid[k] & big[k] for k in big}
            optim.step()
            losses.append(float(loss))
            lrs.append(optim._optimizer.param_groups[0]["lr"])
            o_loss, _, _ = O.train_step(sd_o, opt, batch["feats"], tokens, batch["category"], labels, batch["tgt_length"],
                                        st_o, lr=lr_o)
            check(o_loss, loss.detach(), 5e-5, "trajectory loss %d/%d" % (ep, it))
            k = "b%d." % (ep * steps + it)
            for kk, v in batch.items():
                if kk == "feats":
                    for i, f in enumerate(v):
                        out[k + "feats%d" % i] = f
                else:
                    out[k + kk] = v
        names, info = crit.get_loss_info()
        infos.append(info)
        optim.epoch_update_learning_rate()
        lr_o = max(opt["minimum_learning_rate"], opt["decay"] * lr_o)
        assert abs(lr_o - optim.get_lr()) < 1e-12
    worst = 0.0
    for k, v in model.state_dict().items():
        out["after." + k] = v
        if k in solid:
            out["solid." + k] = solid[k]
            if solid[k].any():
                worst = max(worst, float((sd_o[k] - v).abs()[solid[k]].max()))
    assert worst < 3e-4, worst
    out.update(losses=np.array(losses), lrs=np.array(lrs), loss_names=np.array(names),
               loss_info=np.array(infos, dtype=np.float64), final_lr=optim.get_lr())
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **npz(out))
    print(f"[{name}] ok  losses {losses[0]:.4f} -> {losses[-1]:.4f}  oracle-vs-reference weights {worst:.2e}")


def host_case():
    """SURVEY 8f rows 3-4, host side: what the reference's k-best queue (misc/logger.py:81-211), text helpers
    (misc/utils.py:21-147) and Python caption scorers (coco-caption/pycocoevalcap/{bleu,rouge,cider}) return on seeded
    inputs.  Stored as JSON (inputs + expected outputs)."""
    import shutil
    import tempfile
    os.chdir(REF)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from misc.logger import k_PriorityQueue
    from misc.utils import analyze_length_novel_unique, duplicate, to_sentence
    sys.path.insert(0, os.path.join(REF, "coco-caption"))
    from pycocoevalcap.bleu.bleu import Bleu
    from pycocoevalcap.cider.cider import Cider
    from pycocoevalcap.rouge.rouge import Rouge
    os.chdir(ROOT)
    rs = np.random.RandomState(7)
    words = ["a", "man", "woman", "is", "are", "playing", "cooking", "the", "guitar", "dog", "running", "on", "in", "stage",
             "kitchen", "and", "singing", "with", "cat", "ball", "two", "people", "talking", "car", "driving"]

    def sentence(lo, hi):
        return " ".join(words[i] for i in rs.randint(0, len(words), size=rs.randint(lo, hi)))
    out = {}
    # --- metrics
    gts, res = {}, {}
    for v in range(14):
        vid = "video%d" % v
        gts[vid] = [sentence(3, 11) for _ in range(rs.randint(2, 6))]
        base = gts[vid][0].split(" ")
        hyp = [w if rs.rand() > 0.3 else words[rs.randint(len(words))] for w in base][:rs.randint(2, len(base) + 1)]
        res[vid] = [" ".join(hyp)]
    res["video3"] = [gts["video3"][1]]                      # an exact match
    res["video5"] = ["zebra xylophone"]                     # nothing in common
    with contextlib.redirect_stdout(io.StringIO()):
        b, b_each = Bleu(4).compute_score(gts, res)
        r, r_each = Rouge().compute_score(gts, res)
        c, c_each = Cider().compute_score(gts, res)
    out["metrics"] = dict(gts=gts, res=res, bleu=list(b), bleu_each=[list(x) for x in b_each], rouge=float(r),
                          rouge_each=[float(x) for x in r_each], cider=float(c), cider_each=[float(x) for x in c_each])
    # --- text helpers
    dups = []
    for _ in range(60):
        w = sentence(4, 12).split(" ")
        for _ in range(rs.randint(0, 3)):                   # plant adjacent / one-apart repeats
            n = rs.randint(1, 4)
            i = rs.randint(0, max(1, len(w) - n))
            gap = [words[rs.randint(len(words))]] if rs.rand() < 0.4 else []
            w = w[:i + n] + gap + w[i:i + n] + w[i + n:]
        s_in = " ".join(w)
        s_out, rep = duplicate(s_in)
        dups.append([s_in, s_out, rep])
    out["duplicate"] = dups
    vocab = {i: w for i, w in enumerate(["<pad>", "<unk>", "<bos>", "<eos>", "<mask>", "<vis>"] + words)}
    hyps = [[int(x) for x in rs.randint(0, len(vocab), size=9)] for _ in range(20)]
    out["to_sentence"] = [[h, to_sentence(h, vocab)] for h in hyps]
    gt_data = {"video%d" % v: [[2] + [int(x) for x in rs.randint(6, len(vocab), size=rs.randint(2, 6))] + [3] for _ in range(3)]
               for v in range(8)}
    preds = {"video%d" % v: [{"caption": " ".join(vocab[w] for w in gt_data["video%d" % (v % 8)][v % 3][1:-1]) if v % 2 else sentence(2, 7)}]
             for v in range(12)}
    splits = {"train": [0, 1, 2, 3, 4], "validate": [5, 6], "test": [7]}
    a = analyze_length_novel_unique(gt_data, preds, vocab, splits, n=1)
    out["analyze"] = dict(gt_data=gt_data, preds=preds, splits=splits, vocab={str(k): v for k, v in vocab.items()},
                          ave_length=a[0], novel=a[1], unique=a[2], usage=a[3], grams=a[4], gram4=a[5])
    # --- k-best queue
    runs = []
    for k_best, standard, tol in ((1, ["METEOR", "CIDEr"], 3), (3, ["METEOR", "CIDEr"], 4), (2, ["CIDEr"], 2)):
        tmp = tempfile.mkdtemp()
        try:
            opt = {"checkpoint_path": tmp, "tolerence": tol}
            folder = os.path.join(tmp, "tmp_models")
            q = k_PriorityQueue(k_best_model=k_best, folder_path=folder, standard=standard)
            trace, base = [], 0.2
            for ep in range(14):
                base += 0.03 if ep < 5 else -0.004
                res_ep = {"Bleu_4": float(base + 0.02 * rs.rand()), "METEOR": float(0.5 * base + 0.02 * rs.rand()),
                          "ROUGE_L": float(base + 0.1 + 0.02 * rs.rand()), "CIDEr": float(1.5 * base + 0.05 * rs.rand()),
                          "epoch": ep}
                if ep in (7, 8):
                    res_ep["CIDEr"] = trace[6]["res"]["CIDEr"]         # ties
                given = dict(res_ep)
                with open(os.path.join(tmp, "checkpoint.pth.tar"), "w") as f:
                    f.write("epoch %d" % ep)
                name = "model_%04d.pth.tar" % ep
                ok, info = q.check(res_ep, opt, os.path.join(folder, name), name)
                kept = sorted(os.listdir(folder)) if k_best > 1 else []
                best_file = open(os.path.join(tmp, "best.pth.tar")).read() if k_best == 1 else None
                trace.append(dict(res=given, ok=bool(ok), info=info, failed=q.continuous_failed_count, kept=kept,
                                  best_file=best_file, sum=float(res_ep["Sum"]), best_epoch=q.best_res.get("epoch", -1),
                                  qsize=q.qsize()))
                if not ok:
                    break
            runs.append(dict(k_best=k_best, standard=standard, tolerence=tol, trace=trace))
        finally:
            shutil.rmtree(tmp)
    out["kbest"] = runs
    with open(os.path.join(GOLD, "tiny_host.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("[host] ok: %d videos scored, %d duplicate cases, %d k-best runs (%s steps)" % (
        len(gts), len(dups), len(runs), [len(r["trace"]) for r in runs]))


def goldlen_case():
    """opt['load_generated_captions'] = True (decoding/na_generate.py:25-26,118-122): the length beam is centred on the
    lengths of the captions handed to translate_batch instead of on the length head's top-k"""
    decode_case("tiny_nacf_goldlen_decode", "NACF", ["-wc"], V=101, B=4, F_=6, gold=True, variants={
        "mp_ct_gold": dict(paradigm="mp", use_ct=True, iterations=5, length_beam_size=6, beam_alpha=1.35,
                           load_generated_captions=True),
        "mp_gold5": dict(paradigm="mp", use_ct=False, iterations=3, length_beam_size=5, beam_alpha=1.0,
                         load_generated_captions=True),
        "ef_gold": dict(paradigm="ef", use_ct=True, q=1, q_iterations=1, length_beam_size=4, beam_alpha=1.0,
                        load_generated_captions=True),
    })


def main():
    torch.manual_seed(0)
    if os.environ.get("ONLY_HOST"):
        host_case()
        return
    if os.environ.get("ONLY_TRAJ"):
        trajectory_case()
        return
    if os.environ.get("ONLY_DATA"):
        data_case()
        return
    if os.environ.get("ONLY_CKPT"):
        checkpoint_case()
        return
    if os.environ.get("ONLY_POS"):
        train_case("tiny_nacf_pos_train", "NACF", ["-wc", "--pos_attention"], V=101, B=3, F_=6)
        return
    if os.environ.get("ONLY_POS_LN"):
        train_case("tiny_nacf_pos_ln_train", "NACF", ["-wc", "--pos_attention", "--with_layernorm"], V=101, B=3, F_=6)
        return
    if os.environ.get("ONLY_LN"):
        train_case("tiny_nacf_ln_train", "NACF", ["-wc", "--with_layernorm", "--norm_type", "ln"], V=101, B=3, F_=6)
        return
    if os.environ.get("ONLY_VARIANTS"):
        train_case("tiny_nab_variants_train", "NAB", ["--num_hidden_layers_decoder", "2", "--hidden_act", "gelu",
                                                      "--enhance_input", "0", "--no_encoder_bn", "-tie"],
                   V=101, B=3, F_=6, dataset="Youtube2Text", beta=(0.0, 1.0))
        return
    if os.environ.get("ONLY_GATE"):
        train_case("tiny_nab_nogate_train", "NAB", [], V=101, B=3, F_=6, dataset="Youtube2Text", beta=(0.0, 1.0), override={"gate": False})
        return
    if os.environ.get("ONLY_PMLM"):
        train_case("tiny_nacf_pmlm_train", "NACF", ["-wc"], V=101, B=3, F_=6, override={"parallel_mlm": True})
        train_case("tiny_nab_pmlm_ln_train", "NAB", ["--with_layernorm"], V=101, B=3, F_=6, dataset="Youtube2Text", beta=(0.0, 1.0),
                   override={"parallel_mlm": True})
        return
    if os.environ.get("ONLY_GOLDLEN"):
        goldlen_case()
        return
    if os.environ.get("ONLY_LWE"):
        train_case("tiny_nacf_lwe_train", "NACF", ["-wc"], V=101, B=3, F_=6, override={"load_word_embeddings": True})
        return
    if os.environ.get("ONLY_WATCH"):
        train_case("tiny_arb_watch_train", "ARB", ["-wc", "--watch", "3"], V=101, B=3, F_=6)
        ar_case("tiny_arb_watch_beam", "ARB", ["-wc", "--watch", "3"], V=101, B=4, F_=6, beam_size=3, topk=1, alpha=1.0,
                eos_boost=EOS_BOOST)
        return
    if os.environ.get("ONLY_AR"):
        ar_case("tiny_arb2_beam", "ARB2", ["-wc"], V=101, B=3, F_=6)
        ar_case("tiny_arb_beam", "ARB", ["-wc"], V=101, B=3, F_=6)
        ar_case("tiny_arb_beam_eos", "ARB", ["-wc"], V=101, B=6, F_=6, beam_size=5, topk=3, alpha=1.0, eos_boost=EOS_BOOST)
        ar_case("tiny_arb2_beam_eos", "ARB2", ["-wc"], V=101, B=6, F_=6, beam_size=4, topk=1, alpha=0.7, eos_boost=EOS_BOOST)
        return
    # training: forward + loss + grads + Adam (tiny, dropout 0)
    train_case("tiny_nacf_train", "NACF", ["-wc"], V=101, B=3, F_=6)
    train_case("tiny_nab_train", "NAB", [], V=101, B=3, F_=6, dataset="Youtube2Text", beta=(0.0, 1.0))
    train_case("tiny_arb2_train", "ARB2", ["-wc"], V=101, B=3, F_=6)
    train_case("tiny_arb_train", "ARB", ["-wc"], V=101, B=3, F_=6)
    # option variants the reference runs: 2 layers, erf-gelu, no mean-pooled input, no encoder BN, tied weights
    train_case("tiny_nab_variants_train", "NAB", ["--num_hidden_layers_decoder", "2", "--hidden_act", "gelu",
                                                  "--enhance_input", "0", "--no_encoder_bn", "-tie"],
               V=101, B=3, F_=6, dataset="Youtube2Text", beta=(0.0, 1.0))
    train_case("tiny_nacf_ln_train", "NACF", ["-wc", "--with_layernorm", "--norm_type", "ln"], V=101, B=3, F_=6)
    train_case("tiny_nacf_pos_train", "NACF", ["-wc", "--pos_attention"], V=101, B=3, F_=6)
    train_case("tiny_nacf_pos_ln_train", "NACF", ["-wc", "--pos_attention", "--with_layernorm"], V=101, B=3, F_=6)
    # opt['gate'] = False (models/Encoder.py:10-25,64): HighWay without its gate, out = x + tanh(w1 x)
    train_case("tiny_nab_nogate_train", "NAB", [], V=101, B=3, F_=6, dataset="Youtube2Text", beta=(0.0, 1.0), override={"gate": False})
    # opt['parallel_mlm'] = True (models/bert.py:253-254): the self-attention block of every layer without its residual
    train_case("tiny_nacf_pmlm_train", "NACF", ["-wc"], V=101, B=3, F_=6, override={"parallel_mlm": True})
    train_case("tiny_nab_pmlm_ln_train", "NAB", ["--with_layernorm"], V=101, B=3, F_=6, dataset="Youtube2Text", beta=(0.0, 1.0),
               override={"parallel_mlm": True})
    # opt['load_word_embeddings'] = True (models/bert.py:51-53,77-79): a 768-wide word table and its projection to dim_hidden
    train_case("tiny_nacf_lwe_train", "NACF", ["-wc"], V=101, B=3, F_=6, override={"load_word_embeddings": True})
    # --watch 3 (opts.py:32): AR self-attention sees the last three tokens only (models/Decoder.py:23-39)
    train_case("tiny_arb_watch_train", "ARB", ["-wc", "--watch", "3"], V=101, B=3, F_=6)
    ar_case("tiny_arb_watch_beam", "ARB", ["-wc", "--watch", "3"], V=101, B=4, F_=6, beam_size=3, topk=1, alpha=1.0,
            eos_boost=EOS_BOOST)
    checkpoint_case()
    data_case()
    trajectory_case()
    host_case()
    # NA decode: all paradigms, +-ct, per-iteration tokens/probs
    decode_case("tiny_nacf_decode", "NACF", ["-wc"], V=101, B=4, F_=6, variants={
        "mp_ct": dict(paradigm="mp", use_ct=True, iterations=5, length_beam_size=6, beam_alpha=1.35),
        "mp": dict(paradigm="mp", use_ct=False, iterations=5, length_beam_size=6, beam_alpha=1.0),
        "mp_lbs3": dict(paradigm="mp", use_ct=True, iterations=3, length_beam_size=3, beam_alpha=1.35),
        "ef_ct": dict(paradigm="ef", use_ct=True, q=1, q_iterations=1, length_beam_size=6, beam_alpha=1.35),
        "l2r_ct": dict(paradigm="l2r", use_ct=True, q=1, q_iterations=1, length_beam_size=6, beam_alpha=1.35),
        "ef_q2": dict(paradigm="ef", use_ct=False, q=2, q_iterations=2, length_beam_size=4, beam_alpha=1.0),
        "l2r_q2": dict(paradigm="l2r", use_ct=False, q=2, q_iterations=2, length_beam_size=4, beam_alpha=1.0),
    })
    decode_case("tiny_nab_decode", "NAB", [], V=101, B=4, F_=6, dataset="Youtube2Text", variants={
        "mp": dict(paradigm="mp", use_ct=False, iterations=5, length_beam_size=5, beam_alpha=1.0),
    })
    decode_case("tiny_nacf_teacher", "NACF", ["-wc"], V=101, B=3, F_=6, teacher_method="ARB", variants={
        "mp_ct": dict(paradigm="mp", use_ct=True, iterations=5, length_beam_size=6, beam_alpha=1.35),
        "mp_md": dict(paradigm="mp", use_ct=True, iterations=4, length_beam_size=4, beam_alpha=1.35,
                      masking_decision=True),
    })
    goldlen_case()
    ar_case("tiny_arb2_beam", "ARB2", ["-wc"], V=101, B=3, F_=6)
    ar_case("tiny_arb_beam", "ARB", ["-wc"], V=101, B=3, F_=6)
    ar_case("tiny_arb_beam_eos", "ARB", ["-wc"], V=101, B=6, F_=6, beam_size=5, topk=3, alpha=1.0, eos_boost=EOS_BOOST)
    ar_case("tiny_arb2_beam_eos", "ARB2", ["-wc"], V=101, B=6, F_=6, beam_size=4, topk=1, alpha=0.7, eos_boost=EOS_BOOST)
    full_shape_case()


def check_fixtures():
    """python oracle/make_golden.py --check: regenerate every fixture into a scratch directory and compare it with the committed
    one array by array (npz), value by value (json) and tensor by tensor (the checkpoint); exit status 1 on ANY difference.  This is
    the one-command pin of the oracle's fixtures against the reference (needs /root/reference: this container only)."""
    global GOLD
    import tempfile
    committed = GOLD
    with tempfile.TemporaryDirectory() as tmp:
        GOLD = tmp
        main()
        GOLD = committed
        bad, n_arrays = [], 0
        names = sorted(set(os.listdir(tmp)) | {f for f in os.listdir(committed) if f.endswith((".npz", ".json", ".pth.tar"))})
        for name in names:
            a, b = os.path.join(tmp, name), os.path.join(committed, name)
            if not (os.path.exists(a) and os.path.exists(b)):
                bad.append("%s: %s" % (name, "not committed" if os.path.exists(a) else "not regenerated"))
                continue
            if name.endswith(".npz"):
                x, y = np.load(a, allow_pickle=False), np.load(b, allow_pickle=False)
                for k in sorted(set(x.files) | set(y.files)):
                    n_arrays += 1
                    if k not in x.files or k not in y.files:
                        bad.append("%s[%s]: only in the %s file" % (name, k, "regenerated" if k in x.files else "committed"))
                    elif x[k].dtype != y[k].dtype or x[k].shape != y[k].shape or x[k].tobytes() != y[k].tobytes():
                        bad.append("%s[%s]: differs" % (name, k))
            elif name.endswith(".json"):
                n_arrays += 1
                if json.load(open(a)) != json.load(open(b)):
                    bad.append("%s: differs" % name)
            else:
                x, y = torch.load(a, map_location="cpu", weights_only=True), torch.load(b, map_location="cpu", weights_only=True)

                def same(u, v):
                    if isinstance(u, dict):
                        return isinstance(v, dict) and u.keys() == v.keys() and all(same(u[k], v[k]) for k in u)
                    if torch.is_tensor(u):
                        return torch.is_tensor(v) and u.dtype == v.dtype and torch.equal(u, v)
                    return u == v
                n_arrays += 1
                if not same(x, y):
                    bad.append("%s: differs" % name)
    print("make_golden --check: %d files, %d arrays compared, %d differ" % (len(names), n_arrays, len(bad)))
    for b in bad:
        print("  DIFF", b)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    if "--check" in sys.argv[1:]:
        sys.argv = [sys.argv[0]]
        check_fixtures()
    elif len(sys.argv) > 1:
        sys.exit("usage: python oracle/make_golden.py [--check]      (ONLY_* environment variables select a single case)")
    else:
        main()
