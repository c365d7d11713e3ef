"""Deterministic synthetic weights and batches of the shapes BASELINE.json names
(SURVEY.md section 8d): input generators shared by bench.py, smoke() and the tests.
Nothing here is compute of the captioning path.

`init_state_dict` is a build-defined seeded initialiser (NOT torch's module
init order): every tensor is drawn from a generator seeded by (seed, crc32(name))
so any machine regenerates any tensor independently -- the golden fixtures of
the full-shape case store only the seed and per-tensor checksums.
"""
import math
from typing import Dict, Tuple

import torch

from .config.Constants import BOS, EOS, MASK, PAD, VIS
from .opts import complete_opt as full_opt

SD = Dict[str, torch.Tensor]


def decoder_prefix(opt: dict) -> str:
    """`decoder.bert.` for BertDecoderDisentangled, `decoder.` for BertDecoder (state_dict contract)"""
    return "decoder.bert." if opt["decoder"] == "BertDecoderDisentangled" else "decoder."


def param_shapes(opt: dict) -> Dict[str, Tuple[int, ...]]:
    """Names/shapes of the reference state_dict (SURVEY.md section 8a parameter table)."""
    opt = full_opt(opt)
    d, V, ml = opt["dim_hidden"], opt["vocab_size"], opt["max_len"]
    ff = opt["intermediate_size"]
    sh: Dict[str, Tuple[int, ...]] = {}
    for ch in opt["modality"].lower():
        n = f"encoder.Encoder_{ch.upper()}."
        sh[n + "0.weight"] = (d, opt["dim_" + ch]); sh[n + "0.bias"] = (d,)
        for w in (("w1", "w2") if opt.get("gate", True) else ("w1",)):     # HighWay(with_gate=False) has no w2 (models/Encoder.py:13-15)
            sh[n + f"1.{w}.weight"] = (d, d); sh[n + f"1.{w}.bias"] = (d,)
    if not opt["no_encoder_bn"]:
        for i in range(len(opt["modality"])):
            if opt.get("norm_type", "bn").lower() == "bn":
                n = f"joint_representation_learner.bn{i}."
                for k in ("weight", "bias", "running_mean", "running_var"):
                    sh[n + k] = (d,)
                sh[n + "num_batches_tracked"] = ()
            else:
                n = f"joint_representation_learner.ln{i}."
                sh[n + "weight"] = (d,); sh[n + "bias"] = (d,)
    if "length" in opt["crit"]:
        n = "auxiliary_task_predictor.layers.0.net."
        sh[n + "0.weight"] = (d, d); sh[n + "0.bias"] = (d,)
        sh[n + "3.weight"] = (ml, d); sh[n + "3.bias"] = (ml,)
    p = decoder_prefix(opt)
    if opt.get("load_word_embeddings", False):       # models/bert.py:51-53: a 768-wide table and its projection
        sh[p + "embedding.word_embeddings.weight"] = (V, 768)
        sh[p + "embedding.word_embeddings_prj.weight"] = (d, 768); sh[p + "embedding.word_embeddings_prj.bias"] = (d,)
    else:
        sh[p + "embedding.word_embeddings.weight"] = (V, d)
    sh[p + "embedding.position_embeddings.weight"] = (ml, d)
    if opt["with_category"]:
        sh[p + "embedding.category_embeddings.weight"] = (opt["num_category"], d)
    sh[p + "embedding.LayerNorm.weight"] = (d,); sh[p + "embedding.LayerNorm.bias"] = (d,)
    pos_att = bool(opt.get("pos_attention", False))
    if pos_att:                                  # models/bert.py:63-65
        sh[p + "embedding.pos_LN.weight"] = (d,); sh[p + "embedding.pos_LN.bias"] = (d,)
    for i in range(opt["num_hidden_layers_decoder"]):
        l = f"{p}layer.{i}."
        for a in (("attention", "pos_attention", "attend_to_enc_output") if pos_att else ("attention", "attend_to_enc_output")):
            for q in ("query", "key", "value"):
                sh[f"{l}{a}.self.{q}.weight"] = (d, d); sh[f"{l}{a}.self.{q}.bias"] = (d,)
            sh[f"{l}{a}.output.dense.weight"] = (d, d); sh[f"{l}{a}.output.dense.bias"] = (d,)
        sh[l + "intermediate.dense.weight"] = (ff, d); sh[l + "intermediate.dense.bias"] = (ff,)
        sh[l + "output.dense.weight"] = (d, ff); sh[l + "output.dense.bias"] = (d,)
        if opt.get("with_layernorm", False):
            for m in (("attention.output", "pos_attention.output") if pos_att else ("attention.output",)) + \
                     ("attend_to_enc_output.output", "output"):
                sh[f"{l}{m}.LayerNorm.weight"] = (d,); sh[f"{l}{m}.LayerNorm.bias"] = (d,)
    sh["tgt_word_prj.weight"] = (V, d)
    if opt.get("tie_weights", False):      # models/seq2seq.py:30-33: shared with the word embedding + a bias
        sh["tgt_word_prj.bias"] = (V,)
    return sh


def init_state_dict(opt: dict, seed: int = 0) -> SD:
    """Build-defined seeded initialiser (NOT torch's module init order): each
    tensor is drawn U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (nn.Linear's default
    bound), embeddings N(0,1) with the PAD row zeroed, norm weights 1 / biases
    0, from a generator seeded by (seed, crc32(name)) so any tensor can be
    regenerated independently on any machine."""
    import zlib
    sd: SD = {}
    for name, shape in param_shapes(opt).items():
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.zeros((), dtype=torch.long)
        elif name.endswith("running_mean"):
            sd[name] = torch.zeros(shape)
        elif name.endswith("running_var"):
            sd[name] = torch.ones(shape)
        elif "embedding.LayerNorm.weight" in name or (".bn" in name and name.endswith("weight")):
            sd[name] = torch.ones(shape)
        elif "embedding.LayerNorm.bias" in name or (".bn" in name and name.endswith("bias")):
            sd[name] = torch.zeros(shape)
        elif "LayerNorm.weight" in name or "pos_LN.weight" in name or (".ln" in name and name.endswith("weight")):
            sd[name] = 1.0 + 0.2 * (torch.rand(shape, generator=g) * 2 - 1)      # optional LayerNorms: non-trivial affine
        elif "LayerNorm.bias" in name or "pos_LN.bias" in name or (".ln" in name and name.endswith("bias")):
            sd[name] = 0.1 * (torch.rand(shape, generator=g) * 2 - 1)
        elif "embeddings.weight" in name:
            w = torch.randn(shape, generator=g)
            if "word_embeddings" in name:
                w[PAD].zero_()
            sd[name] = w
        else:
            fan_in = shape[-1] if len(shape) == 2 else None
            if fan_in is None:  # bias: bound from the matching weight's fan_in
                wname = name[:-4] + "weight"
                fan_in = param_shapes(opt)[wname][1]
            bound = 1.0 / math.sqrt(fan_in)
            sd[name] = (torch.rand(shape, generator=g) * 2 - 1) * bound
    if full_opt(opt).get("tie_weights", False):
        sd["tgt_word_prj.weight"] = sd[decoder_prefix(full_opt(opt)) + "embedding.word_embeddings.weight"]
    return sd


def synth_batch(opt: dict, B: int, n_frames: int, seed: int = 1, beta=(0.35, 0.9)) -> dict:
    """Synthetic batch per SURVEY.md section 8d: feats U[0,1) per modality, category,
    MLM inputs/labels as dataloader.py:349-381, visual-word inputs/labels as
    dataloader.py:383-425 (>=1 real word per row here), tgt_length rows sum to 1."""
    opt = full_opt(opt)
    g = torch.Generator().manual_seed(seed)
    L, V = opt["max_len"], opt["vocab_size"]
    feats = [torch.rand(B, n_frames, opt["dim_" + ch], generator=g) for ch in opt["modality"].lower()]
    category = torch.randint(0, opt["num_category"], (B, 1), generator=g)
    lens = torch.randint(4, L, (B,), generator=g)
    gold = torch.randint(6, V, (B, L), generator=g)
    pos = torch.arange(L).unsqueeze(0)
    valid = pos < lens.unsqueeze(1)
    gold = gold * valid
    tokens = gold.clone(); labels = torch.zeros_like(gold)
    tokens_1 = torch.where(valid, torch.full_like(gold, VIS), torch.zeros_like(gold))
    labels_1 = torch.where(valid, torch.full_like(gold, MASK), torch.zeros_like(gold))
    for b in range(B):
        n = int(lens[b])
        lo = max(int(n * beta[0]), 1); hi = max(int(n * beta[1]), 1)
        if hi == lo:
            hi += 1
        k = int(torch.randint(lo, hi, (1,), generator=g))
        ind = torch.randperm(n, generator=g)[:k]
        tokens[b, ind] = MASK
        labels[b, ind] = gold[b, ind]
        kv = max(1, int(round(0.4 * n)))
        indv = torch.randperm(n, generator=g)[:kv]
        labels_1[b, indv] = gold[b, indv]
    tgt_length = torch.zeros(B, L)
    tgt_length[torch.arange(B), lens] = 1.0
    out = dict(feats=feats, category=category, tokens=tokens, labels=labels,
               tokens_1=tokens_1, labels_1=labels_1, tgt_length=tgt_length, lens=lens, gold=gold)
    if opt["decoding_type"] == "ARFormer":
        # AR: <bos> w ... <eos> PAD..., labels = tokens shifted by one (run.py:70-82)
        ar = torch.zeros(B, L, dtype=torch.long)
        l1 = torch.zeros(B, L, dtype=torch.long)
        for b in range(B):
            n = min(int(lens[b]), L - 2)
            ar[b, 0] = BOS; ar[b, 1:1 + n] = gold[b, :n]; ar[b, 1 + n] = EOS
            # dataloader.py:417-419: [<bos>] + (<mask> | visual word) * n + [<eos>]
            l1[b, 0] = BOS; l1[b, 1:1 + n] = labels_1[b, :n]; l1[b, 1 + n] = EOS
        out["tokens"] = ar; out["labels"] = ar[:, 1:].clone()
        out["tokens_1"] = torch.where(ar.ne(PAD), torch.full_like(ar, VIS), ar)
        out["labels_1"] = l1[:, 1:].clone()
    return out
