"""CPU oracle for the NACF hot path -- TEST INFRASTRUCTURE ONLY.

This module is the parity checker, not the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product path (``nacf_amd``) never routes through it and fails
loudly when the HIP library is missing.

What it is: a plain eager PyTorch fp32 *functional* restatement of the
reference's algorithm for the path named in BASELINE.json (``north_star``),
operating on a flat ``dict[str, Tensor]`` keyed exactly like the reference
``state_dict`` (SURVEY.md section 8a "Parameter table").  Every function cites the
reference file:line it follows (paths relative to the upstream checkout).

Pinning: the reference has no tests or golden vectors of its own (SURVEY.md
section 4), so this oracle is pinned by outputs of the reference itself, imported in
the build container by ``oracle/make_golden.py``; the resulting small vectors
live in ``tests/golden/*.npz`` and ``tests/test_oracle_golden.py`` re-checks the
oracle against them on every run (CPU, no reference needed).

Floating point is fp32 everywhere (the reference has no AMP); token ids,
labels and categories are int64.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

PAD, UNK, BOS, EOS, MASK, VIS = 0, 1, 2, 3, 4, 5  # config/Constants.py:1-6

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------
# option handling
# --------------------------------------------------------------------------
DEFAULT_OPT = dict(  # opts.py:23-50,116-129 (defaults that shape the model)
    modality="mi", dim_i=2048, dim_m=2048, dim_a=1, dim_o=1, dim_hidden=512,
    num_hidden_layers_decoder=1, num_attention_heads=8, intermediate_size=2048,
    hidden_act="gelu_new", hidden_dropout_prob=0.5,
    attention_probs_dropout_prob=0.0, max_len=30, layer_norm_eps=1e-5, watch=0,
    pos_attention=False, enhance_input=2, with_layernorm=False,
    with_category=False, num_category=20, encoder_dropout=0.5,
    no_encoder_bn=False, norm_type="bn", tie_weights=False,
    fusion="temporal_concat", crit=["lang"], nv_weights=[0.8, 1.0],
    visual_word_generation=False, decoding_type="ARFormer",
    decoder="BertDecoder", encoder="Encoder_HighWay", gate=True, parallel_mlm=False, load_word_embeddings=False,
)


def full_opt(opt: dict) -> dict:
    o = dict(DEFAULT_OPT)
    o.update(opt)
    return o


def decoder_prefix(opt: dict) -> str:
    """`decoder.bert.` for BertDecoderDisentangled, `decoder.` for BertDecoder
    (models/Decoder.py:181-186; SURVEY.md section 8a parameter table)."""
    return "decoder.bert." if opt["decoder"] == "BertDecoderDisentangled" else "decoder."


# --------------------------------------------------------------------------
# activations
# --------------------------------------------------------------------------
def gelu_new(x: Tensor) -> Tensor:
    """tanh-form GELU, models/bert.py:12-13."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x * x * x)))


def gelu_erf(x: Tensor) -> Tensor:
    """models/bert.py:9-10."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


ACT = {"gelu_new": gelu_new, "gelu": gelu_erf, "relu": F.relu,
       "swish": lambda x: x * torch.sigmoid(x)}


def _dropout(x: Tensor, p: float, training: bool) -> Tensor:
    # The oracle is only ever compared at p == 0 or in eval mode: RNG streams of
    # a CPU and a GPU implementation cannot match (SURVEY.md section 7 "hard parts").
    return F.dropout(x, p, training) if (training and p > 0) else x


# --------------------------------------------------------------------------
# encoder side: SURVEY.md section 8a rows 1-4
# --------------------------------------------------------------------------
def encoder_stream(sd: SD, name: str, x: Tensor, p: float, training: bool, gate: bool = True) -> Tensor:
    """One modality of Encoder_HighWay: Linear -> HighWay -> Dropout.
    models/Encoder.py:9-25 (HighWay), :62-66 (Sequential)."""
    h = F.linear(x, sd[f"encoder.{name}.0.weight"], sd[f"encoder.{name}.0.bias"])
    y = torch.tanh(F.linear(h, sd[f"encoder.{name}.1.w1.weight"], sd[f"encoder.{name}.1.w1.bias"]))
    if not gate:                                                        # models/Encoder.py:24-25 (opt['gate'] = False)
        return _dropout(h + y, p, training)
    g = torch.sigmoid(F.linear(h, sd[f"encoder.{name}.1.w2.weight"], sd[f"encoder.{name}.1.w2.bias"]))
    out = g * h + (1.0 - g) * y
    return _dropout(out, p, training)


def batchnorm_rows(sd: SD, name: str, x: Tensor, training: bool,
                   new_stats: Optional[dict] = None, eps: float = 1e-5,
                   momentum: float = 0.1) -> Tensor:
    """nn.BatchNorm1d over the flattened B*F rows,
    models/joint_representation.py:40-45.  Train: batch statistics (biased
    variance for normalisation, unbiased for the running update)."""
    B, T, D = x.shape
    flat = x.reshape(B * T, D)
    w, b = sd[f"{name}.weight"], sd[f"{name}.bias"]
    if training:
        mean = flat.mean(0)
        var_b = flat.var(0, unbiased=False)
        if new_stats is not None:
            n = flat.shape[0]
            var_u = var_b * (n / max(n - 1, 1))
            new_stats[f"{name}.running_mean"] = (1 - momentum) * sd[f"{name}.running_mean"] + momentum * mean.detach()
            new_stats[f"{name}.running_var"] = (1 - momentum) * sd[f"{name}.running_var"] + momentum * var_u.detach()
            new_stats[f"{name}.num_batches_tracked"] = sd[f"{name}.num_batches_tracked"] + 1
    else:
        mean, var_b = sd[f"{name}.running_mean"], sd[f"{name}.running_var"]
    y = (flat - mean) / torch.sqrt(var_b + eps) * w + b
    return y.reshape(B, T, D)


def encode(sd: SD, opt: dict, feats: Sequence[Tensor], training: bool = False,
           new_stats: Optional[dict] = None) -> Dict[str, Tensor]:
    """Seq2Seq.encode, models/seq2seq.py:35-63: per-modality encoder
    (models/Encoder.py:47-59), fusion (models/joint_representation.py:24-53),
    length predictor (models/Predictor.py:23-30)."""
    opt = full_opt(opt)
    modality = opt["modality"].lower()
    assert len(feats) == len(modality)
    outs, hiddens = [], []
    for ch, x in zip(modality, feats):
        o = encoder_stream(sd, f"Encoder_{ch.upper()}", x, opt["encoder_dropout"], training, gate=opt.get("gate", True))
        outs.append(o)
        hiddens.append(o.mean(1))                      # models/Encoder.py:51
    enc_hidden = torch.stack(hiddens, 0).mean(0)       # joint_representation.py:27
    fusion = opt["fusion"]
    if fusion != "none" and not opt["no_encoder_bn"]:
        for i in range(len(outs)):
            if opt["norm_type"].lower() == "bn":
                outs[i] = batchnorm_rows(sd, f"joint_representation_learner.bn{i}", outs[i], training, new_stats)
            else:
                outs[i] = F.layer_norm(outs[i], (outs[i].shape[-1],),
                                       sd[f"joint_representation_learner.ln{i}.weight"],
                                       sd[f"joint_representation_learner.ln{i}.bias"], 1e-5)
    assert fusion in ("temporal_concat", "none"), "fusion='addition' asserts upstream (SURVEY 8a row 3)"
    enc_output = torch.cat(outs, dim=1)
    res = {"enc_output": enc_output, "enc_hidden": enc_hidden}
    if "length" in opt["crit"]:
        res["pred_length"] = length_head(sd, opt, enc_output, training)
    return res


def length_head(sd: SD, opt: dict, enc_output: Tensor, training: bool) -> Tensor:
    """Predictor_length, models/Predictor.py:12-30."""
    pfx = "auxiliary_task_predictor.layers.0.net."
    h = F.relu(F.linear(enc_output.mean(1), sd[pfx + "0.weight"], sd[pfx + "0.bias"]))
    h = _dropout(h, opt["hidden_dropout_prob"], training)
    return torch.log_softmax(F.linear(h, sd[pfx + "3.weight"], sd[pfx + "3.bias"]), dim=-1)


# --------------------------------------------------------------------------
# decoder: SURVEY.md section 8a rows 5-12
# --------------------------------------------------------------------------
def embeddings(sd: SD, pfx: str, opt: dict, ids: Tensor, category: Optional[Tensor],
               additional: Optional[Tensor], training: bool, return_pos: bool = False):
    """BertEmbeddings.forward, models/bert.py:70-108 (return_pos: also the normalised position embeddings, :97-108)."""
    L = ids.shape[1]
    # nn.Embedding(padding_idx=PAD): the PAD row never receives gradient from the lookup (bert.py:53-56)
    e = F.embedding(ids, sd[pfx + "embedding.word_embeddings.weight"], padding_idx=PAD)
    if opt.get("load_word_embeddings", False):                          # bert.py:51-53,77-79: 768-wide rows, projected
        e = F.linear(e, sd[pfx + "embedding.word_embeddings_prj.weight"], sd[pfx + "embedding.word_embeddings_prj.bias"])
    e = e + sd[pfx + "embedding.position_embeddings.weight"][:L].unsqueeze(0)
    if opt["with_category"]:
        e = e + sd[pfx + "embedding.category_embeddings.weight"][category.reshape(-1)].unsqueeze(1)
    if additional is not None:
        e = e + additional
    e = F.layer_norm(e, (e.shape[-1],), sd[pfx + "embedding.LayerNorm.weight"],
                     sd[pfx + "embedding.LayerNorm.bias"], opt["layer_norm_eps"])
    e = _dropout(e, opt["hidden_dropout_prob"], training)
    if not return_pos:
        return e
    pos = sd[pfx + "embedding.position_embeddings.weight"][:L].unsqueeze(0).expand(ids.shape[0], -1, -1)
    pos = F.layer_norm(pos, (pos.shape[-1],), sd[pfx + "embedding.pos_LN.weight"], sd[pfx + "embedding.pos_LN.bias"],
                       opt["layer_norm_eps"])
    return e, _dropout(pos, opt["hidden_dropout_prob"], training)


def mha(sd: SD, pfx: str, opt: dict, q_in: Tensor, kv_in: Tensor, mask: Optional[Tensor],
        training: bool, v_in: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """BertSelfAttention.forward, models/bert.py:139-179.  ``mask`` is
    bool [B, Lq, Lk], True = masked (filled with -10e6, *not* -inf, :161);
    the 1/sqrt(d_k) scale is applied after QK^T (:157-158)."""
    H = opt["num_attention_heads"]
    B, Lq, D = q_in.shape
    Lk = kv_in.shape[1]
    dk = D // H
    q = F.linear(q_in, sd[pfx + "self.query.weight"], sd[pfx + "self.query.bias"]).view(B, Lq, H, dk).permute(0, 2, 1, 3)
    k = F.linear(kv_in, sd[pfx + "self.key.weight"], sd[pfx + "self.key.bias"]).view(B, Lk, H, dk).permute(0, 2, 1, 3)
    v = F.linear(kv_in if v_in is None else v_in, sd[pfx + "self.value.weight"],
                 sd[pfx + "self.value.bias"]).view(B, Lk, H, dk).permute(0, 2, 1, 3)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dk)
    if mask is not None:
        s = s.masked_fill(mask.unsqueeze(1), -10e6)
    p = torch.softmax(s, dim=-1)
    p = _dropout(p, opt["attention_probs_dropout_prob"], training)
    o = torch.matmul(p, v).permute(0, 2, 1, 3).reshape(B, Lq, D)
    return o, p.permute(1, 0, 2, 3)   # probs as [H, B, Lq, Lk] (:179)


def attention_block(sd: SD, pfx: str, opt: dict, q_in: Tensor, kv_in: Tensor,
                    mask: Optional[Tensor], training: bool, v_in: Optional[Tensor] = None,
                    with_residual: bool = True) -> Tuple[Tensor, Tensor]:
    """BertAttention = BertSelfAttention + BertSelfOutput,
    models/bert.py:182-215: dense -> dropout -> + query input (-> LN if enabled).  v_in: a separate value input
    (the position attention: queries and keys from the position embeddings, values from the hidden states).
    with_residual=False: BertSelfOutput gets no input_tensor (bert.py:195-196,212; the parallel_mlm self-attention)."""
    o, p = mha(sd, pfx, opt, q_in, kv_in, mask, training, v_in)
    o = F.linear(o, sd[pfx + "output.dense.weight"], sd[pfx + "output.dense.bias"])
    o = _dropout(o, opt["hidden_dropout_prob"], training)
    if with_residual:
        o = o + q_in
    if opt["with_layernorm"]:
        o = F.layer_norm(o, (o.shape[-1],), sd[pfx + "output.LayerNorm.weight"],
                         sd[pfx + "output.LayerNorm.bias"], opt["layer_norm_eps"])
    return o, p


def bert_layer(sd: SD, pfx: str, opt: dict, x: Tensor, non_pad: Tensor, self_mask: Tensor,
               enc_output: Tensor, training: bool, pos: Optional[Tensor] = None):
    """BertLayer.forward, models/bert.py:262-303."""
    a, p_self = attention_block(sd, pfx + "attention.", opt, x, x, self_mask, training,
                                with_residual=not opt.get("parallel_mlm", False))      # bert.py:253-254
    a = a * non_pad
    if pos is not None:     # bert.py:274-281: q = k = position embeddings, v = hidden; the residual is the query input
        a, _ = attention_block(sd, pfx + "pos_attention.", opt, pos, pos, self_mask, training, v_in=a)
        a = a * non_pad
    c, p_cross = attention_block(sd, pfx + "attend_to_enc_output.", opt, a, enc_output, None, training)
    c = c * non_pad
    u = ACT[opt["hidden_act"]](F.linear(c, sd[pfx + "intermediate.dense.weight"], sd[pfx + "intermediate.dense.bias"]))
    y = F.linear(u, sd[pfx + "output.dense.weight"], sd[pfx + "output.dense.bias"])
    y = _dropout(y, opt["hidden_dropout_prob"], training) + c          # bert.py:241-243
    if opt["with_layernorm"]:
        y = F.layer_norm(y, (y.shape[-1],), sd[pfx + "output.LayerNorm.weight"],
                         sd[pfx + "output.LayerNorm.bias"], opt["layer_norm_eps"])
    y = _dropout(y, opt["hidden_dropout_prob"], training)              # second dropout, bert.py:247
    y = y * non_pad
    embs = y.sum(1) / non_pad.sum(1)                                    # bert.py:301
    return y, embs, (p_self, p_cross)


def decoder_forward(sd: SD, opt: dict, tgt_seq: Tensor, enc_output: Tensor, category: Optional[Tensor],
                    decoding_type: Optional[str] = None, training: bool = False,
                    prefix: Optional[str] = None):
    """BertDecoder.forward, models/Decoder.py:96-178.  Returns (hidden, embs, attentions)."""
    opt = full_opt(opt)
    pfx = decoder_prefix(opt) if prefix is None else prefix
    dtype_ = decoding_type or opt["decoding_type"]
    B, L = tgt_seq.shape
    key_pad = tgt_seq.eq(PAD).unsqueeze(1).expand(-1, L, -1)            # Decoder.py:13-22
    if dtype_ == "NARFormer":
        self_mask = key_pad                                             # Decoder.py:105-107
    else:
        sub = torch.triu(torch.ones(L, L, dtype=torch.bool), diagonal=1)  # Decoder.py:24-39
        w = int(opt["watch"])
        if w != 0 and L >= w:                                           # --watch: only the last `watch` tokens are visible
            assert w > 0
            sub = sub | torch.tril(torch.ones(L, L, dtype=torch.bool), diagonal=-w)
        self_mask = key_pad | sub.unsqueeze(0)
    non_pad = tgt_seq.ne(PAD).float().unsqueeze(-1)                     # Decoder.py:9-11
    additional = None
    if dtype_ == "NARFormer":
        if opt["enhance_input"] == 2:                                   # Decoder.py:136-137
            additional = enc_output.mean(1).unsqueeze(1).expand(-1, L, -1)
        else:
            assert opt["enhance_input"] == 0, "enhance_input=1 crashes upstream (SURVEY 8a row 11)"
    pos = None
    if opt["pos_attention"]:                                            # Decoder.py:144-146: no additional feats here
        h, pos = embeddings(sd, pfx, opt, tgt_seq, category, None, training, return_pos=True)
    else:
        h = embeddings(sd, pfx, opt, tgt_seq, category, additional, training)
    attns = []
    embs = None
    for i in range(opt["num_hidden_layers_decoder"]):
        h, embs, att = bert_layer(sd, f"{pfx}layer.{i}.", opt, h, non_pad, self_mask, enc_output, training, pos)
        attns.append(att)
    return h, embs, attns


def vocab_logits(sd: SD, opt: dict, hidden: Tensor) -> Tensor:
    """tgt_word_prj, models/__init__.py:83 (bias only with tie_weights, seq2seq.py:30-33)."""
    b = sd.get("tgt_word_prj.bias", None)
    return F.linear(hidden, sd["tgt_word_prj.weight"], b)


def forward_train(sd: SD, opt: dict, feats: Sequence[Tensor], tgt_tokens, category: Tensor,
                  training: bool = True, new_stats: Optional[dict] = None) -> Dict[str, object]:
    """Seq2Seq.forward_NARFormer / forward_ARFormer, models/seq2seq.py:86-140;
    NACF/ARB2 run two decoder passes on the same memory (models/Decoder.py:201-215)."""
    opt = full_opt(opt)
    res = encode(sd, opt, feats, training, new_stats)
    passes = list(tgt_tokens) if isinstance(tgt_tokens, (list, tuple)) else [tgt_tokens]
    if opt["decoding_type"] == "ARFormer":
        passes = [t[:, :-1] for t in passes]                            # seq2seq.py:120
    logprobs = []
    for t in passes:
        h, _, _ = decoder_forward(sd, opt, t, res["enc_output"], category, training=training)
        logprobs.append(torch.log_softmax(vocab_logits(sd, opt, h), dim=-1))
    res["tgt_word_logprobs"] = logprobs
    return res


# --------------------------------------------------------------------------
# loss: SURVEY.md section 8a row 14
# --------------------------------------------------------------------------
def criterion(opt: dict, results: dict, labels, tgt_length: Optional[Tensor]):
    """misc/crit.py:21-48 (CritBase), :62-114 (LanguageGeneration), :223
    (legacy nn.KLDivLoss() == mean over all B*max_len elements).
    Returns (total, dict of parts and meters)."""
    opt = full_opt(opt)
    logps: List[Tensor] = results["tgt_word_logprobs"]
    labels = list(labels) if isinstance(labels, (list, tuple)) else [labels] * len(logps)
    weights = opt["nv_weights"] if opt["visual_word_generation"] else [1.0] * len(logps)
    B = logps[0].shape[0]
    lang = 0.0
    info = {}
    for i, (w, lp, lab) in enumerate(zip(weights, logps, labels)):
        nll = -lp.gather(2, lab.unsqueeze(2)).squeeze(2)
        m = lab.ne(PAD)
        lang = lang + w * (nll * m.float()).sum() / B                   # token-SUM / batch, crit.py:40-46,82
        ind = m & lab.ne(MASK) if (i == 0 and opt["visual_word_generation"]) else m
        info[f"acc_hit{i}"] = int((lp.argmax(-1)[ind] == lab[ind]).sum())
        info[f"acc_cnt{i}"] = int(ind.sum())
        if not (i == 0 and opt["visual_word_generation"]):
            info["ppl_sum_logp"] = float((lp.gather(2, lab.unsqueeze(2)).squeeze(2) * m).sum())
            info["ppl_cnt"] = float(m.sum())
    total = lang
    info["lang"] = lang
    if "length" in opt["crit"]:
        pl = results["pred_length"]
        # KLDivLoss(reduction='mean'): sum(t*(log t - x)) / numel, 0*log0 := 0
        t = tgt_length
        kl = torch.where(t > 0, t * (torch.log(t.clamp_min(1e-45)) - pl), torch.zeros_like(pl))
        length = kl.sum() / pl.numel()
        info["length"] = length
        total = total + length
    return total, info


# --------------------------------------------------------------------------
# optimiser step: SURVEY.md section 8a row 15
# --------------------------------------------------------------------------
def adam_step(params: SD, grads: SD, state: dict, lr: float, grad_clip: float = 5.0,
              weight_decay: float = 5e-4, betas=(0.9, 0.999), eps: float = 1e-8) -> None:
    """clip_grad_value_(+-grad_clip) (misc/run.py:260) then torch.optim.Adam with
    L2-in-gradient weight decay on *every* parameter (misc/optim.py:61-62).
    In place on ``params``; ``state`` holds step/m/v."""
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    b1, b2 = betas
    for k, p in params.items():
        g = grads[k].clamp(-grad_clip, grad_clip)
        g = g + weight_decay * p
        m = state.setdefault("m." + k, torch.zeros_like(p))
        v = state.setdefault("v." + k, torch.zeros_like(p))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** t
        bc2 = 1 - b2 ** t
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / bc1)


def trainable_keys(sd: SD) -> List[str]:
    return [k for k, v in sd.items() if v.is_floating_point()
            and not k.endswith("running_mean") and not k.endswith("running_var")]


def train_step(sd: SD, opt: dict, feats, tgt_tokens, category, labels, tgt_length,
               state: dict, lr: float, training: bool = True):
    """One iteration of the hot loop misc/run.py:254-261 on the oracle:
    forward, loss, backward, clip, Adam.  Mutates ``sd`` in place; returns
    (loss, info, grads)."""
    opt = full_opt(opt)
    keys = trainable_keys(sd)
    wkey = decoder_prefix(opt) + "embedding.word_embeddings.weight"
    tied = bool(opt.get("tie_weights", False))
    if tied:  # ONE parameter serves the embedding and the projection (models/seq2seq.py:27-33)
        keys = [k for k in keys if k != "tgt_word_prj.weight"]
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in keys}
    work = dict(sd)
    work.update(leaves)
    if tied:
        work["tgt_word_prj.weight"] = leaves[wkey]
    new_stats: dict = {}
    res = forward_train(work, opt, feats, tgt_tokens, category, training, new_stats)
    loss, info = criterion(opt, res, labels, tgt_length)
    loss.backward()
    grads = {k: (leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(leaves[k])) for k in keys}
    params = {k: sd[k] for k in keys}
    with torch.no_grad():
        adam_step(params, grads, state, lr, opt.get("grad_clip", 5.0), opt.get("weight_decay", 5e-4))
        for k, v in new_stats.items():
            sd[k] = v
        if tied:
            sd["tgt_word_prj.weight"] = sd[wkey]
    return loss.detach(), info, grads


# --------------------------------------------------------------------------
# NA decoding: SURVEY.md section 8a rows 16-23
# --------------------------------------------------------------------------
def enlarge(x: Tensor, k: int) -> Tensor:
    """misc/utils.py:205-213: row b*k+j is a copy of row b."""
    return x.unsqueeze(1).repeat(1, k, *([1] * (x.dim() - 1))).view(x.shape[0] * k, *x.shape[1:])


def predict_length_beam(pred_length: Tensor, lbs: int, length_bias: int, max_len: int,
                        gold_tokens: Optional[Tensor] = None) -> Tensor:
    """decoding/na_generate.py:116-135.  gold_tokens (opt['load_generated_captions'], :25-26,118-122): lbs
    consecutive lengths around the number of non-<pad> tokens of the given captions (no length_bias)."""
    if gold_tokens is not None:
        gold = gold_tokens.ne(PAD).sum(-1)
        beam = (gold - (lbs - 1) // 2).unsqueeze(1) + torch.arange(lbs).unsqueeze(0)
    else:
        beam = pred_length.topk(lbs, dim=1)[1] + length_bias
    beam = beam.clamp(min=4, max=max_len - 1)
    return beam


def na_step(sd: SD, opt: dict, tokens: Tensor, enc_output: Tensor, category: Tensor, pad_mask: Tensor):
    """generate_non_autoregressive + generate_step_with_prob,
    decoding/algorithms.py:7-15,143-167: one NA decoder pass, softmax over V,
    (max, argmax); PAD slots forced to (PAD, 1.0)."""
    h, _, _ = decoder_forward(sd, opt, tokens, enc_output, category, decoding_type="NARFormer")
    probs = torch.softmax(vocab_logits(sd, opt, h), dim=-1)
    mp, idx = probs.max(dim=-1)
    idx = idx.masked_fill(pad_mask, PAD)
    mp = mp.masked_fill(pad_mask, 1.0)
    return idx, mp


def select_worst(token_probs: Tensor, num_mask: Tensor) -> Tensor:
    """decoding/algorithms.py:206-215: per row, the max(1, n) least confident slots."""
    m = torch.zeros_like(token_probs, dtype=torch.bool)
    for i in range(token_probs.shape[0]):
        ind = token_probs[i].topk(max(1, int(num_mask[i])), largest=False, sorted=False)[1]
        m[i, ind] = True
    return m


def teacher_scores(teacher, tokens: Tensor, pad_mask: Tensor, is_last: bool, opt: dict) -> Tensor:
    """scoring_by_teacher, decoding/algorithms.py:169-204.  ``teacher`` is
    None or a tuple (sd_t, opt_t, enc_output_t, category_t) of an AR model."""
    ones = torch.ones(tokens.shape, dtype=torch.float32)
    if teacher is None:
        return ones
    if is_last and opt.get("no_candidate_decision", False):
        return ones
    if (not is_last) and not opt.get("masking_decision", False):
        return ones
    sd_t, opt_t, enc_t, cat_t = teacher
    with_bos = torch.cat([torch.full((tokens.shape[0], 1), BOS, dtype=tokens.dtype), tokens], dim=1)
    h, _, _ = decoder_forward(sd_t, opt_t, with_bos[:, :-1], enc_t, cat_t, decoding_type="ARFormer")
    probs = torch.softmax(vocab_logits(sd_t, opt_t, h), dim=-1)
    probs = probs.gather(2, tokens.unsqueeze(2)).squeeze(2)
    return probs.masked_fill(pad_mask, 1.0)


def _ct_pass(sd, opt, tokens, enc, cat, pad_mask):
    """get_coarse_grained_templates, decoding/algorithms.py:136-141."""
    tokens = tokens.masked_fill(tokens.eq(MASK), VIS)
    tokens, probs = na_step(sd, opt, tokens, enc, cat, pad_mask)
    probs = probs.masked_fill(tokens.eq(MASK), 0.0)
    return tokens, probs


def mask_predict(sd, opt, dec_opt, tokens, enc, cat, teacher=None, collect=None):
    """MaskPredict.generate, decoding/algorithms.py:224-273."""
    T = dec_opt.get("iterations", 5)
    use_ct = dec_opt.get("use_ct", False)
    pad_mask = tokens.eq(PAD)
    seq_lens = tokens.shape[1] - pad_mask.sum(1)
    if use_ct:
        tokens, probs = _ct_pass(sd, opt, tokens, enc, cat, pad_mask)
        T = T + 1
    else:
        tokens, probs = na_step(sd, opt, tokens, enc, cat, pad_mask)
    if collect is not None:
        collect.append((tokens.clone(), probs.clone()))
    for c in range(1, T):
        tp = teacher_scores(teacher, tokens, pad_mask, False, dec_opt)
        if use_ct and c == 1:
            mask_ind = tokens.eq(MASK)
        else:
            ratio = 1.0 - (c / T)
            num_mask = (seq_lens.float() * ratio).long()
            mask_ind = select_worst(probs * tp, num_mask)
        tokens = tokens.masked_fill(mask_ind, MASK)
        nt, npb = na_step(sd, opt, tokens, enc, cat, pad_mask)
        tokens = torch.where(mask_ind, nt, tokens)
        probs = torch.where(mask_ind, npb, probs)
        if collect is not None:
            collect.append((tokens.clone(), probs.clone()))
    tp = teacher_scores(teacher, tokens, pad_mask, True, dec_opt)
    return tokens, (probs * tp).log()


def _refine(sd, opt, dec_opt, tokens, probs, enc, cat, pad_mask, seq_lens, visual_mask, collect):
    """Shared tail of Left2Right / EasyFirst, decoding/algorithms.py:326-339,400-413."""
    T = dec_opt.get("q_iterations", 1)
    for i in range(T):
        if i == 0 and dec_opt.get("use_ct", False):
            mask_ind = visual_mask
        else:
            ratio = 0.4 * (1.0 - (i / T))
            mask_ind = select_worst(probs, (seq_lens.float() * ratio).long())
        tokens = tokens.masked_fill(mask_ind, MASK)
        nt, npb = na_step(sd, opt, tokens, enc, cat, pad_mask)
        tokens = torch.where(mask_ind, nt, tokens)
        probs = torch.where(mask_ind, npb, probs)
        if collect is not None:
            collect.append((tokens.clone(), probs.clone()))
    return tokens, probs


def left2right(sd, opt, dec_opt, tokens, enc, cat, teacher=None, collect=None):
    """Left2Right.generate, decoding/algorithms.py:275-344."""
    q = dec_opt.get("q", 1)
    pad_mask = tokens.eq(PAD)
    L = tokens.shape[1]
    seq_lens = L - pad_mask.sum(1)
    visual_mask = None
    if dec_opt.get("use_ct", False):
        tokens, probs = _ct_pass(sd, opt, tokens, enc, cat, pad_mask)
        visual_mask = tokens.ne(MASK) & tokens.ne(PAD)
    else:
        probs = torch.zeros(tokens.shape).masked_fill(pad_mask, 1.0)
    if collect is not None:
        collect.append((tokens.clone(), probs.clone()))
    slots = [[j for j in range(int(seq_lens[i])) if tokens[i, j] == MASK] for i in range(tokens.shape[0])]
    for cur in range(0, L, q):
        mask_ind = torch.zeros_like(tokens, dtype=torch.bool)
        for i, s in enumerate(slots):
            for j in s[cur:cur + q]:
                mask_ind[i, j] = True
        if mask_ind.sum() == 0:
            break
        tokens = tokens.masked_fill(mask_ind, MASK)
        nt, npb = na_step(sd, opt, tokens, enc, cat, pad_mask)
        tokens = torch.where(mask_ind, nt, tokens)
        probs = torch.where(mask_ind, npb, probs)
        if collect is not None:
            collect.append((tokens.clone(), probs.clone()))
    tokens, probs = _refine(sd, opt, dec_opt, tokens, probs, enc, cat, pad_mask, seq_lens, visual_mask, collect)
    tp = teacher_scores(teacher, tokens, pad_mask, True, dec_opt)
    return tokens, (probs * tp).log()


def easy_first(sd, opt, dec_opt, tokens, enc, cat, teacher=None, collect=None):
    """EasyFirst.generate, decoding/algorithms.py:346-418."""
    q = dec_opt.get("q", 1)
    pad_mask = tokens.eq(PAD)
    seq_lens = tokens.shape[1] - pad_mask.sum(1)
    visual_mask = None
    if dec_opt.get("use_ct", False):
        tokens, probs = _ct_pass(sd, opt, tokens, enc, cat, pad_mask)
        visual_mask = tokens.ne(MASK) & tokens.ne(PAD)
    else:
        probs = torch.zeros(tokens.shape).masked_fill(pad_mask, 1.0)
    if collect is not None:
        collect.append((tokens.clone(), probs.clone()))
    pre = 0
    while True:
        mask_ind = tokens.eq(MASK)
        remain = int(mask_ind.sum())
        if remain == 0 or pre == remain:
            break
        pre = remain
        nt, npb = na_step(sd, opt, tokens, enc, cat, pad_mask)
        cand = npb.masked_fill(~mask_ind, 0.0)                          # algorithms.py:373 (in place upstream)
        sel = torch.zeros_like(mask_ind)
        left = mask_ind.sum(-1)
        for i in range(tokens.shape[0]):
            if int(left[i]) > 0:
                ind = cand[i].topk(min(q, int(left[i])), largest=True, sorted=False)[1]
                sel[i, ind] = True
        tokens = torch.where(sel, nt, tokens)
        probs = torch.where(sel, cand, probs)
        if collect is not None:
            collect.append((tokens.clone(), probs.clone()))
    tokens, probs = _refine(sd, opt, dec_opt, tokens, probs, enc, cat, pad_mask, seq_lens, visual_mask, collect)
    tp = teacher_scores(teacher, tokens, pad_mask, True, dec_opt)
    return tokens, (probs * tp).log()


ALGORITHMS = {"mp": mask_predict, "l2r": left2right, "ef": easy_first}


def generate(sd: SD, opt: dict, dec_opt: dict, enc_res: Dict[str, Tensor], category: Tensor,
             teacher=None, collect: Optional[list] = None, gold_tokens: Optional[Tensor] = None):
    """decoding.generate, decoding/na_generate.py:14-77.  ``dec_opt`` carries
    paradigm / use_ct / iterations / length_beam_size / beam_alpha / q /
    q_iterations / length_bias.  Returns (hypotheses [B, L'], all candidates
    [B, lbs, L'], candidate log-probs [B, lbs, L'], beam lengths [B, lbs])."""
    opt = full_opt(opt)
    lbs = dec_opt.get("length_beam_size", 6)
    alpha = dec_opt.get("beam_alpha", 1.0)
    pred_length = enc_res["pred_length"]
    B = pred_length.shape[0]
    beam = predict_length_beam(pred_length, lbs, dec_opt.get("length_bias", 0), opt["max_len"],
                               gold_tokens if dec_opt.get("load_generated_captions", False) else None)
    Lp = int(beam.max())
    pos = torch.arange(Lp).view(1, 1, Lp)
    is_pad = pos >= beam.unsqueeze(-1)                                  # na_generate.py:39-40
    start = torch.full_like(is_pad, MASK, dtype=torch.long)
    if gold_tokens is not None and dec_opt.get("load_generated_captions", False):
        g = gold_tokens[:, :Lp].clone()                                 # na_generate.py:42-45: the canvas starts from the
        g[g == PAD] = MASK                                              # given captions, <pad> -> <mask>
        start = g.unsqueeze(1).repeat(1, lbs, 1)
    tokens = torch.where(is_pad, torch.full_like(is_pad, PAD, dtype=torch.long), start).view(B * lbs, Lp)
    enc = enlarge(enc_res["enc_output"], lbs)
    cat = enlarge(category, lbs)
    if teacher is not None:
        sd_t, opt_t, enc_t = teacher
        teacher = (sd_t, full_opt(opt_t), enlarge(enc_t, lbs), cat)
    algo = ALGORITHMS[dec_opt.get("paradigm", "mp")]
    hyp, lprobs = algo(sd, opt, dec_opt, tokens, enc, cat, teacher, collect)
    hyp = hyp.view(B, lbs, Lp)
    lprobs = lprobs.view(B, lbs, Lp)
    score = lprobs.sum(-1) / (beam.float() ** alpha)                    # na_generate.py:72
    best = score.max(-1)[1]
    out = hyp.gather(1, best.view(B, 1, 1).expand(-1, 1, Lp)).squeeze(1)
    return out, hyp, lprobs, beam


# --------------------------------------------------------------------------
# AR beam search (config-5 comparator): SURVEY.md section 8a row 24
# --------------------------------------------------------------------------
def ar_beam_search(sd: SD, opt: dict, enc_res: Dict[str, Tensor], category: Tensor,
                   beam_size: int = 5, alpha: float = 1.0, topk: int = 1):
    """Translator.translate_batch_ARFormer + Beam, models/Translator.py:94-161,
    models/Beam.py:5-169, restated per instance (no batching tricks): whole-prefix
    decoder recompute each step, flat top-k over beam x vocab, finished beams
    collected when an EOS is *emitted*, stop when `beam_size` finished or at max_len.
    Returns (list[B] of list[topk] of token lists, list[B] of scores)."""
    opt = full_opt(opt)
    max_len = opt["max_len"]
    B = enc_res["enc_output"].shape[0]
    all_h, all_s = [], []
    n_best = topk  # shrinks monotonically ACROSS instances upstream (Translator.py:84-92): kept as is
    for b in range(B):
        enc = enc_res["enc_output"][b:b + 1].expand(beam_size, -1, -1)
        cat = category[b:b + 1].expand(beam_size, -1)
        scores = torch.zeros(beam_size)
        ys = [torch.full((beam_size,), PAD, dtype=torch.long)]
        ys[0][0] = BOS
        prev: List[Tensor] = []
        finished: List[list] = []
        done = False
        want = max(beam_size, topk)
        for step in range(1, max_len):
            # current hypotheses, sorted by score as Beam.get_tentative_hypothesis does
            if len(ys) == 1:
                seq = ys[0].unsqueeze(1)
            else:
                order = torch.sort(scores, 0, True)[1]
                hyps = []
                for k in order.tolist():
                    h, kk = [], k
                    for j in range(len(prev) - 1, -1, -1):
                        h.append(int(ys[j + 1][kk]))
                        kk = int(prev[j][kk])
                    hyps.append([BOS] + h[::-1])
                seq = torch.tensor(hyps, dtype=torch.long)
            hdn, _, _ = decoder_forward(sd, opt, seq, enc, cat, decoding_type="ARFormer")
            lp = torch.log_softmax(vocab_logits(sd, opt, hdn[:, -1, :]), dim=1)
            V = lp.shape[1]
            if prev:
                lk = lp + scores.unsqueeze(1)
                lk[ys[-1] == EOS] = -1e20
            else:
                lk = lp[0]
            best, bid = lk.reshape(-1).topk(beam_size, 0, True, True)
            scores = best
            pk = bid // V
            prev.append(pk)
            ys.append(bid - pk * V)
            for i in range(beam_size):
                if int(ys[-1][i]) == EOS:
                    finished.append([float(scores[i]), len(ys) - 1, i])
                    if len(finished) >= want:
                        done = True
                        break
            if done:
                break
            if len(ys) == max_len:
                if not finished:
                    for i in range(beam_size):
                        finished.append([float(scores[i]), len(ys) - 1, i])
                break
        for it in finished:
            it[0] /= it[1] ** alpha
        finished.sort(key=lambda a: -a[0])
        hyps, scs = [], []
        n_best = min(n_best, len(finished))
        for sc, t, k in finished[:n_best]:
            h = []
            for j in range(t - 1, -1, -1):
                h.append(int(ys[j + 1][k]))
                k = int(prev[j][k])
            hyps.append(h[::-1])
            scs.append(sc)
        all_h.append(hyps)
        all_s.append(scs)
    return all_h, all_s


# --------------------------------------------------------------------------
# deterministic synthetic model / batch: these are INPUT generators, not part of the
# checked algorithm; they live in the package (nacf_amd.synthetic) so that bench.py
# and the product never import this oracle to make their inputs.  Re-exported here
# for the tests and oracle/make_golden.py.
# --------------------------------------------------------------------------
import nacf_amd  # noqa: E402,F401  (repo-root import alias of the package directory)
from nacf_amd.synthetic import init_state_dict, param_shapes, synth_batch  # noqa: E402,F401
