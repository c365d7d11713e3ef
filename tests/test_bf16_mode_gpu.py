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
    assert last.startswith("gemm_bf16_group_kernel<128, 128, 1, 1, 1,") or "gemm_bf16_kernel<" in last, last
    assert model.flat.images is not None and model.flat.images.ns == 1
    assert optim._optimizer.exp_avg.dtype == torch.float32
    sd = model.state_dict()
    assert all(v.dtype == torch.float32 for v in sd.values() if v.is_floating_point())
    # the images follow the masters: a forward after the last update reads images of exactly the current weights
    model.eval()
    with torch.no_grad():
        e1 = model.encode(feats=batch["feats"])["enc_output"].clone()
        model.flat.images.img.zero_()                        # corrupt the cache ...
        e2 = model.encode(feats=batch["feats"])["enc_output"]   # ... the forward entry rebuilds it
    assert torch.equal(e1, e2)
