"""BERT-style decoder blocks (reference: models/bert.py:46-303).

The classes keep the reference's module / parameter names so checkpoints load
unchanged, but they are parameter holders: `BertLayer.run` issues the layer as
eight launches-with-fused-epilogues on the MI355X:

  qkv   = x Wqkv^T + b                      (one packed [3D, D] MFMA GEMM)
  att   = softmax(QK^T/sqrt(dk) masked) V   (LDS-resident per (row, head))
  a     = (dropout(att Wo^T + b) + x) * non_pad          (fused epilogue)
  q     = a Wq^T + b ;  kv = memory Wkv^T + b (once per VIDEO, shared by both
          NACF passes / all length candidates / all decode iterations)
  c     = (dropout(cross(q, kv) Wo^T + b) + a) * non_pad
  u     = gelu_new(c W1^T + b)                            (fused epilogue)
  y     = dropout(dropout(u W2^T + b) + c) * non_pad      (fused epilogue; the
          reference's BertOutput really applies dropout twice, bert.py:242,247)

Reference quirks kept: mask fill is -1e7 (not -inf); scaling happens after
QK^T; PAD *queries* are not masked, their rows are zeroed by non_pad_mask.
"""
import torch
import torch.nn as nn

from ..config import Constants
from ..runtime import lib as L
from ..runtime import ops
from ..runtime.functional import (CrossAttentionFn, EmbedLNFn, EmbedLNTableFn, LayerNormFn, LinearFn, Pack,
                                  PositionRowsFn, ProjectTableFn, QKVAttentionFn, SelfAttentionFn)


class BertEmbeddings(nn.Module):
    """word + position (+ category) embeddings and LayerNorm (bert.py:46-67)"""

    def __init__(self, config):
        super().__init__()
        # load_word_embeddings (bert.py:51-53): a 768-wide table (e.g. pretrained word vectors) and its projection
        self.projected = bool(getattr(config, 'load_word_embeddings', False))
        if self.projected:
            self.word_embeddings = nn.Embedding(config.vocab_size, 768, padding_idx=Constants.PAD)
            self.word_embeddings_prj = nn.Linear(768, config.dim_hidden)
        else:
            self.word_embeddings = nn.Embedding(config.vocab_size, config.dim_hidden, padding_idx=Constants.PAD)
        self.position_embeddings = nn.Embedding(config.max_len, config.dim_hidden)
        self.category_embeddings = nn.Embedding(config.num_category, config.dim_hidden) if config.with_category else None
        self.LayerNorm = nn.LayerNorm(config.dim_hidden, eps=config.layer_norm_eps)
        # pos_attention (opts.py:33): the normalised position embeddings are a second output (bert.py:63-65,97-108)
        self.pos_LN = nn.LayerNorm(config.dim_hidden, eps=config.layer_norm_eps) if config.pos_attention else None
        self.p = config.hidden_dropout_prob
        self.eps = config.layer_norm_eps

    def nacf_groups(self):
        g = [[self.word_embeddings.weight]]
        if self.projected:
            g += [[self.word_embeddings_prj.weight], [self.word_embeddings_prj.bias]]
        g.append([self.position_embeddings.weight])
        if self.category_embeddings is not None:
            g.append([self.category_embeddings.weight])
        g += [[self.LayerNorm.weight], [self.LayerNorm.bias]]
        if self.pos_LN is not None:
            g += [[self.pos_LN.weight], [self.pos_LN.bias]]
        return g

    def nacf_bind(self, flat, rt):
        self._rt = rt
        cat = self.category_embeddings
        self._cfg = dict(word=flat.pack([self.word_embeddings.weight]), pos=flat.pack([self.position_embeddings.weight]),
                         cat=flat.pack([cat.weight]) if cat is not None else None,
                         ln=flat.pack([self.LayerNorm.weight], [self.LayerNorm.bias]), p=self.p, eps=self.eps,
                         salt=rt.next_salt(), train_word=self.word_embeddings.weight.requires_grad)
        if self.pos_LN is not None:
            self._ln_pos = flat.pack([self.pos_LN.weight], [self.pos_LN.bias])
            self._salt_pos = rt.next_salt()
        if self.projected:
            self._prj = flat.pack([self.word_embeddings_prj.weight], [self.word_embeddings_prj.bias], image='both')
        self._params = [p for p in self.parameters()]

    def run_pos(self, tokens, training):
        """pos_dropout(pos_LN(position_embeddings)) for every slot: [R*Lq, D]  (bert.py:105)"""
        rows = PositionRowsFn.apply(dict(pos=self._cfg['pos']), tokens, *self._params)
        cfg = dict(ln=self._ln_pos, eps=self.eps, p=self.p, salt=self._salt_pos, rng=self._rt.rng(tokens.device),
                   training=training)
        return LayerNormFn.apply(rows, cfg, *self._params)

    def run(self, tokens, category, additional, vdiv, vmod, training):
        cfg = dict(self._cfg, vdiv=vdiv, vmod=vmod, training=training, rng=self._rt.rng(tokens.device))
        if self.category_embeddings is not None:
            assert category is not None, 'with_category models need `category`'
            category = category.reshape(-1).contiguous()
        if self.projected:
            # rows of the PROJECTED table are looked up: one [V, 768] x [768, D] GEMM per call instead of one over every token
            table = ProjectTableFn.apply(dict(table=self._cfg['word'], pack=self._prj, train_word=self._cfg['train_word']),
                                         *self._params)
            return EmbedLNTableFn.apply(additional, table, cfg, tokens, category, *self._params)
        return EmbedLNFn.apply(additional, cfg, tokens, category, *self._params)


class _SelfAttnParams(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.query, self.key, self.value = nn.Linear(d, d), nn.Linear(d, d), nn.Linear(d, d)


class _SelfOutputParams(nn.Module):
    def __init__(self, d, with_layernorm=False, eps=1e-5):
        super().__init__()
        self.dense = nn.Linear(d, d)
        self.LayerNorm = nn.LayerNorm(d, eps=eps) if with_layernorm else None


class BertAttention(nn.Module):
    """.self.{query,key,value} + .output.dense (bert.py:115-215)"""

    def __init__(self, config):
        super().__init__()
        if config.dim_hidden % config.num_attention_heads != 0:
            raise ValueError('The hidden size (%d) is not a multiple of the number of attention heads (%d)'
                             % (config.dim_hidden, config.num_attention_heads))
        self.self = _SelfAttnParams(config.dim_hidden)
        self.output = _SelfOutputParams(config.dim_hidden, config.with_layernorm, config.layer_norm_eps)


class _Dense(nn.Module):
    def __init__(self, i, o, with_layernorm=False, eps=1e-5):
        super().__init__()
        self.dense = nn.Linear(i, o)
        if with_layernorm:
            self.LayerNorm = nn.LayerNorm(o, eps=eps)


class BertLayer(nn.Module):
    def __init__(self, config, is_decoder_layer=True):
        super().__init__()
        # parallel_mlm (bert.py:253-254): the self-attention block's BertSelfOutput gets no residual input
        self.self_residual = not getattr(config, 'parallel_mlm', False)
        if config.hidden_act not in L.ACT_BY_NAME:
            raise NotImplementedError('nacf_amd: hidden_act %s is not built' % config.hidden_act)
        self.attention = BertAttention(config)
        # registration order = the reference's (bert.py:254-260): state_dict order and default-init RNG consumption
        self.pos_attention = BertAttention(config) if (config.pos_attention and is_decoder_layer) else None
        self.attend_to_enc_output = BertAttention(config)
        self.intermediate = _Dense(config.dim_hidden, config.intermediate_size)
        self.output = _Dense(config.intermediate_size, config.dim_hidden, config.with_layernorm, config.layer_norm_eps)
        self.with_layernorm = bool(config.with_layernorm)
        self.eps = config.layer_norm_eps
        self.H = config.num_attention_heads
        self.p = config.hidden_dropout_prob
        # attention_probs_dropout_prob (bert.py:135,169; reference default 0): > 0 runs the LDS-tile attention kernels in training
        self.p_att = float(config.attention_probs_dropout_prob)
        self.act = L.ACT_BY_NAME[config.hidden_act]

    def nacf_groups(self):
        a, c = self.attention, self.attend_to_enc_output
        return [
            [a.self.query.weight, a.self.key.weight, a.self.value.weight],
            [a.self.query.bias, a.self.key.bias, a.self.value.bias],
            [a.output.dense.weight], [a.output.dense.bias],
            [c.self.query.weight], [c.self.query.bias],
            [c.self.key.weight, c.self.value.weight], [c.self.key.bias, c.self.value.bias],
            [c.output.dense.weight], [c.output.dense.bias],
            [self.intermediate.dense.weight], [self.intermediate.dense.bias],
            [self.output.dense.weight], [self.output.dense.bias],
        ] + ([[m.LayerNorm.weight] for m in (a.output, c.output, self.output)] +
             [[m.LayerNorm.bias] for m in (a.output, c.output, self.output)] if self.with_layernorm else []) + \
            ([[pa.self.query.weight, pa.self.key.weight], [pa.self.query.bias, pa.self.key.bias],
              [pa.self.value.weight], [pa.self.value.bias], [pa.output.dense.weight], [pa.output.dense.bias]]
             if (pa := self.pos_attention) is not None else []) + \
            ([[pa.output.LayerNorm.weight], [pa.output.LayerNorm.bias]]
             if (pa := self.pos_attention) is not None and self.with_layernorm else [])

    def nacf_bind(self, flat, rt):
        self._rt = rt
        a, c = self.attention, self.attend_to_enc_output
        self._pk = dict(
            qkv=flat.pack([a.self.query.weight, a.self.key.weight, a.self.value.weight],
                          [a.self.query.bias, a.self.key.bias, a.self.value.bias], image='both'),
            so=flat.pack([a.output.dense.weight], [a.output.dense.bias], image='both'),
            cq=flat.pack([c.self.query.weight], [c.self.query.bias], image='both'),
            ckv=flat.pack([c.self.key.weight, c.self.value.weight], [c.self.key.bias, c.self.value.bias], image='both'),
            co=flat.pack([c.output.dense.weight], [c.output.dense.bias], image='both'),
            f1=flat.pack([self.intermediate.dense.weight], [self.intermediate.dense.bias], image='both'),
            f2=flat.pack([self.output.dense.weight], [self.output.dense.bias], image='both'))
        if self.with_layernorm:
            for key, m in (('ln_so', a.output), ('ln_co', c.output), ('ln_f2', self.output)):
                self._pk[key] = flat.pack([m.LayerNorm.weight], [m.LayerNorm.bias])
        self._salts = [rt.next_salt() for _ in range(4)]
        # (drawn only when used: the salts of every other dropout site stay what they are with p_att = 0)
        self._salts_att = [rt.next_salt() for _ in range(3)] if self.p_att > 0.0 else None
        pa = self.pos_attention
        if pa is not None:
            self._pk['pqk'] = flat.pack([pa.self.query.weight, pa.self.key.weight], [pa.self.query.bias, pa.self.key.bias],
                                        image='both')
            self._pk['pv'] = flat.pack([pa.self.value.weight], [pa.self.value.bias], image='both')
            self._pk['po'] = flat.pack([pa.output.dense.weight], [pa.output.dense.bias], image='both')
            if self.with_layernorm:
                self._pk['ln_po'] = flat.pack([pa.output.LayerNorm.weight], [pa.output.LayerNorm.bias])
            self._salt_pos = rt.next_salt()
        self._params = [p for p in self.parameters()]

    def _att_drop(self, which, training, rng):
        """(p, salt, rng) of the probability dropout of attention block `which` (0 self, 1 cross, 2 position), or None"""
        if not (training and self.p_att > 0.0):
            return None
        return (self.p_att, self._salts_att[which], rng)

    def project_memory(self, enc_output):
        """K|V projection of the visual memory: [Bv, M, D] -> [Bv*M, 2D]"""
        Bv, M, D = enc_output.shape
        return LinearFn.apply(enc_output.reshape(Bv * M, D), None, dict(pack=self._pk['ckv']), *self._params)

    def run(self, x2, tokens, causal, memory_kv, M, vdiv, vmod, training, want_probs, rows=None, pos2=None,
            out_rows=None):
        """x2: [R*L, D] hidden rows; tokens: [R, L]; memory_kv: [Bv*M, 2D]; rows: live (non-<pad>)
        slot list -- the row-wise GEMMs skip <pad> slots, whose outputs are exact zeros anyway."""
        R, Lq = tokens.shape
        rng = self._rt.rng(x2.device)
        tok_flat = tokens.reshape(-1)
        P = self._params
        pk = self._pk
        s = self._salts
        if self.with_layernorm:
            return self._run_layernorm(x2, tokens, causal, memory_kv, M, vdiv, vmod, training, want_probs, rows, pos2)
        if self.pos_attention is not None:
            return self._run_pos(x2, pos2, tokens, causal, memory_kv, M, vdiv, vmod, training, want_probs, rows)
        if out_rows is not None and not torch.is_grad_enabled() and not training:
            return self._run_subset(x2, tokens, causal, memory_kv, M, vdiv, vmod, want_probs, rows, out_rows)
        # each block input (x2, a, c) feeds a Linear AND the residual add of the block's output Linear.  The output
        # Linear's backward runs first (it is downstream), parks its residual gradient in h*, and the input Linear's
        # dX GEMM accumulates onto it (LinearFn: res_sink / dx_acc) -- only wired when both gradients will exist.
        link = torch.is_grad_enabled() and x2.requires_grad
        h1, h2, h3 = ({}, {}, {}) if link else (None, None, None)
        # dead (<pad>) rows: zero-filled where something reads EVERY row -- q|k|v and the cross-attention query (the attention
        # cores and their backward), the layer's output; the three buffers in between (a, c, u) are only ever read through
        # the same row list (the next nn.Linear, the residual of the one after, the weight-gradient GEMMs), so they are not
        # filled (fill=False), and FFN2's dX feeds FFN1's row-list epilogue backward (dx_fill=False)
        nf = dict(fill=False) if rows is not None else {}
        qkv = LinearFn.apply(x2, None, dict(pack=pk['qkv'], rows=rows, dx_acc=h1), *P)
        att, p_self = SelfAttentionFn.apply(qkv, tokens, int(causal), self.H, want_probs, self._att_drop(0, training, rng))
        a = LinearFn.apply(att, x2 if self.self_residual else None, dict(pack=pk['so'], p1=self.p, salt1=s[0], row_tokens=tok_flat, rng=rng,
                                         training=training, rows=rows, res_sink=h1, **nf), *P)
        q = LinearFn.apply(a, None, dict(pack=pk['cq'], rows=rows, dx_acc=h2), *P)
        catt, p_cross = CrossAttentionFn.apply(q, memory_kv, self.H, Lq, M, vdiv, vmod, want_probs, self._att_drop(1, training, rng))
        c = LinearFn.apply(catt, a, dict(pack=pk['co'], p1=self.p, salt1=s[1], row_tokens=tok_flat, rng=rng,
                                         training=training, rows=rows, res_sink=h2, **nf), *P)
        u = LinearFn.apply(c, None, dict(pack=pk['f1'], act=self.act, rows=rows, dx_acc=h3, **nf), *P)
        y = LinearFn.apply(u, c, dict(pack=pk['f2'], p1=self.p, salt1=s[2], p2=self.p, salt2=s[3], row_tokens=tok_flat,
                                      rng=rng, training=training, rows=rows, res_sink=h3, dx_fill=rows is None), *P)
        return y, (p_self, p_cross)

    def _run_subset(self, x2, tokens, causal, memory_kv, M, vdiv, vmod, want_probs, rows, out_rows):
        """Inference, last layer: only the slots in `out_rows` are consumed downstream (mask-predict re-predicts just
        the re-masked slots: decoding/algorithms.py:na_pass).  Keys and values are still projected for every live
        slot, but the query projection and everything after the self-attention (output projections, the
        cross-attention, the FFN -- 5/6 of the layer's GEMM flops) run on `out_rows` only.  Each kept row is
        computed by the same instruction sequence as in `run`, so its values are bit-identical; the other rows of
        the returned tensor are zeros."""
        R, Lq = tokens.shape
        D = x2.shape[1]
        P, pk = self._params, self._pk
        tok_flat = tokens.reshape(-1)
        if 'q_only' not in pk:
            full = pk['qkv']
            pk['q_only'] = Pack(full.w[:D], full.b[:D], None, None)
            pk['kv_only'] = Pack(full.w[D:], full.b[D:], None, None)
        sub = dict(row_tokens=tok_flat, training=False, rows=out_rows)
        att = torch.empty(R * Lq, D, dtype=x2.dtype, device=x2.device)
        p_self = torch.empty(self.H, R, Lq, Lq, dtype=x2.dtype, device=x2.device) if want_probs else None
        # inference: nothing walks the dead rows of q, a, cq, c, u backwards, and forward they are read through `out_rows` only
        # (the attention cores compute garbage for the unlisted queries, which nobody gathers): no zero fill -- 150 MB of zero
        # stores per pass at B = 128 with 6 length candidates; k|v (masked keys multiply V by an exact 0) and y are filled
        # (with want_probs the attention maps of EVERY query are handed back: the unlisted queries must then be zeros, not
        #  uninitialised memory -- ADVICE round 4)
        q = LinearFn.apply(x2, None, dict(pack=pk['q_only'], rows=out_rows, fill=bool(want_probs)), *P)
        kv = LinearFn.apply(x2, None, dict(pack=pk['kv_only'], rows=rows), *P)
        ops.attention_fwd(q, kv[:, :D], kv[:, D:], att, tokens, int(causal), p_self, R, self.H, Lq, Lq, D // self.H, 1, R)
        a = LinearFn.apply(att, x2 if self.self_residual else None, dict(pack=pk['so'], fill=False, **sub), *P)
        cq = LinearFn.apply(a, None, dict(pack=pk['cq'], rows=out_rows, fill=bool(want_probs)), *P)
        catt, p_cross = CrossAttentionFn.apply(cq, memory_kv, self.H, Lq, M, vdiv, vmod, want_probs)
        c = LinearFn.apply(catt, a, dict(pack=pk['co'], fill=False, **sub), *P)
        u = LinearFn.apply(c, None, dict(pack=pk['f1'], act=self.act, rows=out_rows, fill=False), *P)
        y = LinearFn.apply(u, c, dict(pack=pk['f2'], **sub), *P)
        return y, (p_self, p_cross)

    def can_run_last(self, causal):
        """_run_last serves the plain layer under the plain triangular mask (no LayerNorm / position-attention variants, no
        --watch window: models/Decoder.py:23-39)"""
        return (not self.with_layernorm) and self.pos_attention is None and int(causal) == 1

    def _run_last(self, x2, tokens, memory_kv, M, vdiv, vmod, rows):
        """Inference, last layer, autoregressive step (models/Translator.py:105-111 reads `dec_output[:, -1, :]` and nothing
        else): the hidden state of the LAST slot of every sequence, [R, D].  Keys and values are projected for the whole
        prefix; the query projection, the two attention cores (one query per sequence: under the triangular mask the last query
        sees every key, so no causal mask is needed), both output projections and the FFN run densely on R rows -- the
        same sums per kept row as `run`, in the order the dense GEMMs' tiles give."""
        R, Lq = tokens.shape
        D = x2.shape[1]
        P, pk = self._params, self._pk
        if 'q_only' not in pk:
            full = pk['qkv']
            pk['q_only'] = Pack(full.w[:D], full.b[:D], None, None)
            pk['kv_only'] = Pack(full.w[D:], full.b[D:], None, None)
        x_last = x2.view(R, Lq, D)[:, Lq - 1, :].contiguous()
        tok_last = tokens[:, Lq - 1].contiguous()
        sub = dict(row_tokens=tok_last, training=False)
        kv = LinearFn.apply(x2, None, dict(pack=pk['kv_only'], rows=rows), *P)
        q = LinearFn.apply(x_last, None, dict(pack=pk['q_only']), *P)
        att = torch.empty(R, D, dtype=x2.dtype, device=x2.device)
        ops.attention_fwd(q, kv[:, :D], kv[:, D:], att, tokens, 0, None, R, self.H, 1, Lq, D // self.H, 1, R)
        a = LinearFn.apply(att, x_last if self.self_residual else None, dict(pack=pk['so'], **sub), *P)
        cq = LinearFn.apply(a, None, dict(pack=pk['cq']), *P)
        catt, _ = CrossAttentionFn.apply(cq, memory_kv, self.H, 1, M, vdiv, vmod, False)
        c = LinearFn.apply(catt, a, dict(pack=pk['co'], **sub), *P)
        u = LinearFn.apply(c, None, dict(pack=pk['f1'], act=self.act), *P)
        return LinearFn.apply(u, c, dict(pack=pk['f2'], **sub), *P)

    def _run_layernorm(self, x2, tokens, causal, memory_kv, M, vdiv, vmod, training, want_probs, rows, pos2=None):
        """with_layernorm=True (opts.py:36): LayerNorm sits between the residual add and the <pad> mask,
        so the GEMM epilogue stops at the residual and a LayerNorm kernel finishes the block:
           a = LN(dropout(dense(att)) + x) * non_pad                  (BertSelfOutput, bert.py:192-200)
           y = dropout(LN(dropout(dense(u)) + c)) * non_pad           (BertOutput, bert.py:240-247)"""
        R, Lq = tokens.shape
        rng = self._rt.rng(x2.device)
        tok_flat = tokens.reshape(-1)
        P, pk, s = self._params, self._pk, self._salts
        ln = lambda key, p=0.0, salt=0: dict(ln=pk[key], eps=self.eps, p=p, salt=salt, rng=rng, training=training,
                                             row_tokens=tok_flat)
        qkv = LinearFn.apply(x2, None, dict(pack=pk['qkv'], rows=rows), *P)
        att, p_self = SelfAttentionFn.apply(qkv, tokens, int(causal), self.H, want_probs, self._att_drop(0, training, rng))
        a = LinearFn.apply(att, x2 if self.self_residual else None, dict(pack=pk['so'], p1=self.p, salt1=s[0], rng=rng, training=training, rows=rows), *P)
        a = LayerNormFn.apply(a, ln('ln_so'), *P)
        if self.pos_attention is not None:      # pos_attention with LayerNorm: the block of _run_pos, LN before the <pad> mask
            assert pos2 is not None, 'pos_attention layers need the position embeddings'
            pqk = LinearFn.apply(pos2, None, dict(pack=pk['pqk'], rows=rows), *P)
            pv = LinearFn.apply(a, None, dict(pack=pk['pv'], rows=rows), *P)
            patt = QKVAttentionFn.apply(pqk, pv, tokens, int(causal), self.H, self._att_drop(2, training, rng))
            a = LinearFn.apply(patt, pos2, dict(pack=pk['po'], p1=self.p, salt1=self._salt_pos, rng=rng, training=training,
                                                rows=rows), *P)
            a = LayerNormFn.apply(a, ln('ln_po'), *P)
        q = LinearFn.apply(a, None, dict(pack=pk['cq'], rows=rows), *P)
        catt, p_cross = CrossAttentionFn.apply(q, memory_kv, self.H, Lq, M, vdiv, vmod, want_probs, self._att_drop(1, training, rng))
        c = LinearFn.apply(catt, a, dict(pack=pk['co'], p1=self.p, salt1=s[1], rng=rng, training=training, rows=rows), *P)
        c = LayerNormFn.apply(c, ln('ln_co'), *P)
        u = LinearFn.apply(c, None, dict(pack=pk['f1'], act=self.act, rows=rows), *P)
        y = LinearFn.apply(u, c, dict(pack=pk['f2'], p1=self.p, salt1=s[2], rng=rng, training=training, rows=rows), *P)
        y = LayerNormFn.apply(y, ln('ln_f2', self.p, s[3]), *P)
        return y, (p_self, p_cross)

    def _run_pos(self, x2, pos2, tokens, causal, memory_kv, M, vdiv, vmod, training, want_probs, rows):
        """pos_attention=True (opts.py:33, bert.py:274-281): between the self- and the cross-attention block sits a
        third attention whose queries AND keys come from the (normalised) position embeddings, whose values come from
        the hidden states, and whose residual is the query input, i.e. the position embeddings:
            a' = (dropout(dense(softmax(Qp Kp^T / sqrt(dk), mask) V(a))) + pos) * non_pad"""
        assert pos2 is not None, 'pos_attention layers need the position embeddings'
        R, Lq = tokens.shape
        rng = self._rt.rng(x2.device)
        tok_flat = tokens.reshape(-1)
        P, pk, s = self._params, self._pk, self._salts
        common = dict(row_tokens=tok_flat, rng=rng, training=training, rows=rows)
        qkv = LinearFn.apply(x2, None, dict(pack=pk['qkv'], rows=rows), *P)
        att, p_self = SelfAttentionFn.apply(qkv, tokens, int(causal), self.H, want_probs, self._att_drop(0, training, rng))
        a = LinearFn.apply(att, x2 if self.self_residual else None, dict(pack=pk['so'], p1=self.p, salt1=s[0], **common), *P)
        pqk = LinearFn.apply(pos2, None, dict(pack=pk['pqk'], rows=rows), *P)
        pv = LinearFn.apply(a, None, dict(pack=pk['pv'], rows=rows), *P)
        patt = QKVAttentionFn.apply(pqk, pv, tokens, int(causal), self.H, self._att_drop(2, training, rng))
        a = LinearFn.apply(patt, pos2, dict(pack=pk['po'], p1=self.p, salt1=self._salt_pos, **common), *P)
        q = LinearFn.apply(a, None, dict(pack=pk['cq'], rows=rows), *P)
        catt, p_cross = CrossAttentionFn.apply(q, memory_kv, self.H, Lq, M, vdiv, vmod, want_probs, self._att_drop(1, training, rng))
        c = LinearFn.apply(catt, a, dict(pack=pk['co'], p1=self.p, salt1=s[1], **common), *P)
        u = LinearFn.apply(c, None, dict(pack=pk['f1'], act=self.act, rows=rows), *P)
        y = LinearFn.apply(u, c, dict(pack=pk['f2'], p1=self.p, salt1=s[2], p2=self.p, salt2=s[3], **common), *P)
        return y, (p_self, p_cross)
