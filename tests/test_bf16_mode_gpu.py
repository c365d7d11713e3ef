"""GPU: the bf16 THROUGHPUT mode (BASELINE.json configs[1]: bf16 compute, fp32 master weights) at model level.

The mode is not a parity mode -- operands are rounded to 8 significant bits -- so nothing here is bit-exact; what is
checked is that it is a sound approximation of the fp32-accurate path on the same weights and batch (SURVEY.md section 7:
"bf16 runs report token agreement separately"), that training in it works end to end under the hipGraph engine, and that
the master weights / optimiser state / checkpoints stay fp32."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(opt, dev, mode):
    import nacf_amd
    from nacf_amd import synthetic as S
    from nacf_amd.runtime import ops
    ops.set_gemm_mode(mode)
    m = nacf_amd.get_model(opt)
    m.load_state_dict(S.init_state_dict(opt, seed=0))
    return m.to(dev)


@pytest.fixture()
def restore_mode():
    from nacf_amd.runtime import ops
    before = ops.gemm_mode()
    yield
    ops.set_gemm_mode(before)


def test_bf16_mode_tracks_the_fp32_accurate_path(dev, restore_mode):
    import nacf_amd
    from nacf_amd import synthetic as S
    from nacf_amd.misc.crit import get_criterion
    from nacf_amd.models.Translator import Translator
    opt = nacf_amd.opts.make_opt("NACF", "MSRVTT", with_category=True, max_len=20, vocab_size=10547, n_frames=60,
                                 fused_loss=True, hidden_dropout_prob=0.0, encoder_dropout=0.0, use_ct=True, iterations=5,
                                 length_beam_size=6, beam_alpha=1.35, paradigm="mp")
    b = S.synth_batch(opt, 32, 60, seed=5)
    out = {}
    for mode in ("bf16x3", "bf16"):
        model = _model(opt, dev, mode)
        model.train()
        crit = get_criterion(model.opt)
        model.zero_grad()
        res = model(feats=[f.to(dev) for f in b["feats"]], tgt_tokens=[b["tokens_1"].to(dev), b["tokens"].to(dev)],
                    category=b["category"].to(dev))
        res["tgt_word_labels"] = [b["labels_1"].to(dev), b["labels"].to(dev)]
        res["tgt_length"] = b["tgt_length"].to(dev)
        loss = crit.get_loss(res)
        loss.backward()
        grad = model.flat.grad.clone()
        assert model.flat.data.dtype == torch.float32 and grad.dtype == torch.float32      # fp32 master weights / gradients
        model.eval()
        with torch.no_grad():
            enc = model.encode(feats=[f.to(dev) for f in b["feats"]])
            hid, *_ = model.decoder(b["tokens"].to(dev), enc_output=enc["enc_output"], category=b["category"].to(dev))
            hid = hid[-1] if isinstance(hid, list) else hid
            logp = model.vocab_logprobs(hid)
            hyp, _ = Translator(model, dict(model.opt, decode_graph="off"), device=dev).translate_batch(enc, b["category"].to(dev), None, None)
        out[mode] = (float(loss.detach()), grad, logp.float(), hyp)
    (l3, g3, p3, h3), (l1, g1, p1, h1) = out["bf16x3"], out["bf16"]
    live = b["tokens"].ne(0).to(dev)
    assert abs(l1 - l3) < 2e-3 * abs(l3), (l1, l3)                                   # same loss to ~3 digits
    cos = float(torch.nn.functional.cosine_similarity(g1, g3, dim=0))
    assert cos > 0.999, cos                                                          # same descent direction
    d = (p1 - p3).abs()[live]
    assert float(d.max()) < 5e-2 and float(d.mean()) < 5e-3, (float(d.max()), float(d.mean()))
    agree = float((p1.argmax(-1) == p3.argmax(-1))[live].float().mean())
    assert agree > 0.97, agree                                                       # teacher-forced argmax agreement
    w = min(h1.shape[1], h3.shape[1])
    assert float((h1[:, :w] == h3[:, :w]).float().mean()) > 0.9                       # free-running NA decode mostly agrees
    print("bf16 vs bf16x3: loss %.5f / %.5f, grad cosine %.6f, logp max err %.2e mean %.2e, argmax agreement %.4f"
          % (l1, l3, cos, float(d.max()), float(d.mean()), agree))


def test_bf16_mode_trains_under_the_graph_engine(dev, restore_mode):
    """BASELINE configs[1] at its own shapes: hipGraph-replayed steps in the throughput mode: the loss falls, every launch was a bf16 kernel, the
    optimiser state is fp32 and the checkpointed weights are the fp32 masters (not the bf16 images)"""
    import nacf_amd
    from nacf_amd import synthetic as S
    from nacf_amd.misc.crit import get_criterion
    from nacf_amd.misc.optim import get_optimizer
    from nacf_amd.misc.run import get_forword_results
    from nacf_amd.runtime import lib as L
    from nacf_amd.runtime.engine import TrainStep
    # BASELINE.json configs[1]: NAB, bf16 compute / fp32 master weights, batch 64, seq_len 20, MSRVTT shapes
    opt = nacf_amd.opts.make_opt("NAB", "MSRVTT", with_category=True, max_len=20, vocab_size=10547, n_frames=60, fused_loss=True,
                                 learning_rate=2e-3)
    model = _model(opt, dev, "bf16")
    model.train()
    crit, optim = get_criterion(model.opt), get_optimizer(model.opt, model)
    b = S.synth_batch(opt, 64, 60, seed=2)
    batch = {"feats": [f.to(dev) for f in b["feats"]], "tokens": b["tokens"].to(dev), "labels": b["labels"].to(dev),
             "category": b["category"].to(dev), "length_target": b["tgt_length"].to(dev)}
    engine = TrainStep(model, crit, optim, lambda bb: get_forword_results(model.opt, model, bb, dev), graph="on")
    engine(batch)
    first = float(engine.loss)
    for _ in range(30):
        engine()
    assert engine.captured and float(engine.loss) < 0.8 * first, (first, float(engine.loss))
    last = L.load().nacf_gemm_last_kernel().decode()      # the step's last GEMM launch: the grouped weight gradients, NS = 1
    assert last.startswith(("g256_dw_group_kernel<1>", "gemm_bf16_group_kernel<128, 128, 1, 1, 1,")) or "gemm_bf16_kernel<" in last, last
    assert model.flat.images is not None and model.flat.images.ns == 1
    assert optim._optimizer.exp_avg.dtype == torch.float32
    sd = model.state_dict()
    assert all(v.dtype == torch.float32 for v in sd.values() if v.is_floating_point())
    # the images follow the masters: a forward after the last update reads images of exactly the current weights (the replayed
    # step bumped FlatParams.version, so the first inference entry rebuilds them; later ones do not touch an up-to-date cache)
    model.eval()
    with torch.no_grad():
        assert model.flat.images_version != model.flat.version
        e1 = model.encode(feats=batch["feats"])["enc_output"].clone()
        assert model.flat.images_version == model.flat.version
        model.flat.images.img.zero_()                        # corrupt the cache and declare the weights written ...
        model.flat.touch()
        e2 = model.encode(feats=batch["feats"])["enc_output"]   # ... the forward entry rebuilds it
    assert torch.equal(e1, e2)


def test_bf16_mode_against_the_oracle_with_the_measured_tolerance(dev, restore_mode):
    """the throughput mode was only ever compared with this repo's own exact mode.  Here: BASELINE configs[1]
    (NAB, MSRVTT shapes, batch 64 -- and the NACF step at 32) directly against the CPU ORACLE (oracle/nacf_oracle.py, pinned to
    the reference by the golden fixtures): loss and every parameter gradient of one training step at dropout 0.
    The mode rounds every GEMM operand to bf16 (8 significant bits), so this is NOT the north_star's fp32 tolerance; the bars
    below are the MEASURED deviation (1x MI355X, this test's output) with about 2x head-room, so that a regression shows up:
      loss: 1.1e-5 / 1.5e-5 relative (NAB B=64 / NACF B=32)                                          -> bar 2e-4
      gradients, max error of a tensor / that tensor's max |g|: median over the tensors 4.6e-3 / 3.7e-3  -> bar 1e-2;
        worst tensor outside the length head 1.6e-2 (an encoder bias, a sum over 3840 rows)            -> bar 4e-2;
        the length head's first Linear 0.18 / 0.33 (weight), 0.06 / 0.12 (bias): its ReLU sits on pre-activations of ~1e-3
        at random init, a bf16-rounded operand flips units on and off (models/Predictor.py:15-20)      -> bar 0.6
      cosine of the whole gradient vector with the oracle's 0.999997 / 0.999999                        -> bar 0.9999"""
    import nacf_amd
    from nacf_amd import synthetic as S
    from nacf_amd.misc.crit import get_criterion
    from oracle import nacf_oracle as O
    for method, B in (("NAB", 64), ("NACF", 32)):
        opt = nacf_amd.opts.make_opt(method, "MSRVTT", with_category=True, max_len=20, vocab_size=10547, n_frames=60,
                                     fused_loss=True, hidden_dropout_prob=0.0, encoder_dropout=0.0)
        sd = S.init_state_dict(opt, seed=0)
        b = S.synth_batch(opt, B, 60, seed=11)
        two = method == "NACF"
        model = _model(opt, dev, "bf16")
        model.train()
        crit = get_criterion(model.opt)
        model.zero_grad()
        res = model(feats=[f.to(dev) for f in b["feats"]],
                    tgt_tokens=[b["tokens_1"].to(dev), b["tokens"].to(dev)] if two else b["tokens"].to(dev), category=b["category"].to(dev))
        res["tgt_word_labels"] = [b["labels_1"].to(dev), b["labels"].to(dev)] if two else b["labels"].to(dev)
        res["tgt_length"] = b["tgt_length"].to(dev)
        loss = crit.get_loss(res)
        loss.backward()
        grads = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()}
        s_ = {k: v.clone() for k, v in sd.items()}
        o_loss, _, o_grads = O.train_step(s_, opt, b["feats"], [b["tokens_1"], b["tokens"]] if two else b["tokens"], b["category"],
                                          [b["labels_1"], b["labels"]] if two else b["labels"], b["tgt_length"], {},
                                          lr=opt["learning_rate"])
        rel_loss = abs(float(loss) - float(o_loss)) / abs(float(o_loss))
        errs, dot, n1, n2 = [], 0.0, 0.0, 0.0
        for k, g in grads.items():
            ref = o_grads[k]
            scale = float(ref.abs().max())
            dot += float((g.double() * ref.double()).sum()); n1 += float((g.double() ** 2).sum()); n2 += float((ref.double() ** 2).sum())
            if scale > 1e-7:
                errs.append((float((g - ref).abs().max()) / scale, k))
        errs.sort(reverse=True)
        cos = dot / (n1 ** 0.5 * n2 ** 0.5)
        med = errs[len(errs) // 2][0]
        print("bf16 mode vs ORACLE, %s B=%d: loss rel err %.2e; gradient error / max|g|: worst %.2e (%s), median %.2e; cosine %.6f"
              % (method, B, rel_loss, errs[0][0], errs[0][1], med, cos))
        print("   three worst tensors:", ["%s %.2e" % (k, e) for e, k in errs[:3]])
        assert rel_loss < 2e-4, rel_loss
        head = [e for e, k in errs if k.startswith("auxiliary_task_predictor")]
        rest = [e for e, k in errs if not k.startswith("auxiliary_task_predictor")]
        assert max(head) < 0.6 and max(rest) < 4e-2 and med < 1e-2, (errs[:4], med)
        assert cos > 0.9999, cos
