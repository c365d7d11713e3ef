"""Transformer decoders (reference: models/Decoder.py:68-215).

`BertDecoder` / `BertDecoderDisentangled` keep the reference call convention
    decoder(tgt_seq, enc_output=..., category=..., [decoding_type=], [output_attentions=])
      -> ([hidden], embs[, attentions])            (BertDecoder)
      -> (hidden | [h_vis, h_mlm], embs[, attns])  (BertDecoderDisentangled)
and add three keyword-only accelerators the NA decoding loop uses:
    row_map=('div', k) | ('mod', B): which video a decoder row belongs to, so
        the visual memory is never physically repeated (replaces enlarge(),
        misc/utils.py:205-213);
    memory_kv=[...]: per-layer K|V projections of the memory (project_memory),
        computed once per video instead of once per pass;
    pooled_memory: mean_t(enc_output) for enhance_input=2;
    out_row_set: (inference) the slots whose hidden state will be read -- the last layer computes only those.
Masks are never materialised: kernels derive key-padding / causal masks from
the token ids (models/Decoder.py:9-39).  The two NACF / ARB2 passes run as ONE
batch of 2B rows that share weights and memory (models/Decoder.py:201-215).
"""
import torch
import torch.nn as nn

from ..runtime import ops
from ..runtime.functional import BatchedPasses, HalvesFn, MeanTimeFn
from .bert import BertEmbeddings, BertLayer

__all__ = ('BertDecoder', 'BertDecoderDisentangled')


class EmptyObject(object):
    pass


def dict2obj(d):
    obj = EmptyObject()
    obj.__dict__.update(d)
    return obj


class BertDecoder(nn.Module):
    def __init__(self, config, embedding=None):
        super().__init__()
        if isinstance(config, dict):
            config = dict2obj(config)
        self.embedding = BertEmbeddings(config) if embedding is None else embedding
        self.layer = nn.ModuleList([BertLayer(config, is_decoder_layer=True)
                                    for _ in range(config.num_hidden_layers_decoder)])
        self.pos_attention = config.pos_attention
        self.enhance_input = config.enhance_input
        if self.enhance_input not in (0, 2):
            # enhance_input=1 (resampling) crashes in the reference itself on torch>=1.2 (Decoder.py:43)
            raise ValueError('enhance_input shoud be either 0 or 2 in nacf_amd')
        self.watch = config.watch
        assert self.watch >= 0, 'watch is a window length (models/Decoder.py:28)'
        self.decoding_type = config.decoding_type
        self.pack_rows = bool(getattr(config, 'pack_rows', True))   # skip <pad> slots in the row-wise GEMMs

    def get_word_embeddings(self):
        return self.embedding.word_embeddings

    def set_word_embeddings(self, we):
        self.embedding.word_embeddings = we

    def nacf_groups(self):
        return self.embedding.nacf_groups() + [g for l in self.layer for g in l.nacf_groups()]

    def nacf_bind(self, flat, rt):
        self.__dict__['_flat'] = flat       # (not a sub-module / parameter: keep it out of nn.Module's bookkeeping)
        self.embedding.nacf_bind(flat, rt)
        for l in self.layer:
            l.nacf_bind(flat, rt)

    def project_memory(self, enc_output):
        self._flat.ensure_images()          # a direct call on cached encoder outputs: the weights may have moved on
        return [l.project_memory(enc_output) for l in self.layer]

    @staticmethod
    def _row_map(row_map, R, Bv):
        if row_map is None:
            if R == Bv:
                return 1, Bv
            assert R % Bv == 0, 'decoder rows (%d) must be a multiple of the memory batch (%d)' % (R, Bv)
            return R // Bv, Bv          # enlarge() semantics: row b*k+j reads video b
        kind, val = row_map
        if kind == 'mod':
            assert val == Bv
            return 1, Bv                # pass-major batching: row p*B+b reads video b
        assert kind == 'div' and R == val * Bv
        return val, Bv

    def forward(self, tgt_seq, enc_output=None, category=None, signals=None, tags=None, **kwargs):
        self._flat.ensure_images()
        decoding_type = kwargs.get('decoding_type', self.decoding_type)
        output_attentions = kwargs.get('output_attentions', False)
        if isinstance(enc_output, list):
            assert len(enc_output) == 1
            enc_output = enc_output[0]
        if decoding_type not in ('NARFormer', 'ARFormer'):
            raise NotImplementedError('nacf_amd: decoding_type %s is not built' % decoding_type)
        tgt_seq = tgt_seq.contiguous()
        R, Lq = tgt_seq.shape
        enc_output = enc_output.contiguous()
        Bv, M, D = enc_output.shape
        vdiv, vmod = self._row_map(kwargs.get('row_map'), R, Bv)
        training = self.training
        additional = None
        if decoding_type == 'NARFormer' and self.enhance_input == 2:
            additional = kwargs.get('pooled_memory')
            if additional is None:
                additional = MeanTimeFn.apply(enc_output)
        if signals is not None:     # Decoder.py:141-142: added to (or standing in for) the additional features of the embedding
            # the embedding kernel broadcasts ONE additional row per video over its slots: per-video signals only
            if signals.numel() != Bv * D or signals.shape[0] != Bv:
                raise NotImplementedError('nacf_amd: `signals` must be one row per video ([B, D] or [B, 1, D]); got %s'
                                          % (tuple(signals.shape),))
            sig = signals.reshape(Bv, D).to(enc_output.dtype)
            additional = sig if additional is None else additional.reshape(Bv, D) + sig
        pos2 = None
        if self.pos_attention:      # Decoder.py:144-146: the embedding gets no additional features in this mode
            additional = None
            pos2 = self.embedding.run_pos(tgt_seq, training)
        hidden = self.embedding.run(tgt_seq, category, additional, vdiv, vmod, training)
        memory_kv = kwargs.get('memory_kv')
        # live-row list of the [R, L] slot grid: by default every non-<pad> slot of tgt_seq; the NA
        # decoding loop passes the list of its initial canvas (a superset that stays valid while
        # slots are re-masked / re-predicted)
        rows = kwargs.get('row_set')
        if rows is None and self.pack_rows:
            rows = ops.rowset_build(tokens=tgt_seq.reshape(-1))
        # inference only: the slots whose hidden state the caller will read (a subset of `rows`); the last layer
        # restricts its query-side work to them and `embs` (a mean over ALL slots) is not produced
        out_rows = kwargs.get('out_row_set') if not (training or torch.is_grad_enabled()) else None
        x2 = hidden.reshape(R * Lq, D)
        all_attentions = ()
        causal = (1 + self.watch) if decoding_type == 'ARFormer' else 0
        # autoregressive step of the beam search (models/Beam.py): only the last slot's hidden state is read
        last_only = bool(kwargs.get('last_slot_only')) and not (training or torch.is_grad_enabled()) and not output_attentions \
            and self.layer[-1].can_run_last(causal)
        for i, layer in enumerate(self.layer):
            kv = memory_kv[i] if memory_kv is not None else layer.project_memory(enc_output)
            last = i == len(self.layer) - 1
            if last and last_only:
                y = layer._run_last(x2, tgt_seq, kv, M, vdiv, vmod, rows)
                return ([y.view(R, 1, D)], None,)
            x2, att = layer.run(x2, tgt_seq, (1 + self.watch) if decoding_type == 'ARFormer' else 0, kv, M, vdiv, vmod, training,
                                output_attentions, rows, pos2, out_rows if last else None)
            if output_attentions:
                all_attentions = all_attentions + (att,)
        hidden = x2.view(R, Lq, D)
        embs = None
        if out_rows is None and kwargs.get('want_embs', True):     # Seq2Seq drops `embs` (seq2seq.py:94,124): it asks not to
            with torch.no_grad():
                embs = ops.masked_mean_fwd(hidden.detach(), tgt_seq, torch.empty(R, D, device=hidden.device))
        outputs = ([hidden], embs,)
        if output_attentions:
            outputs = outputs + (all_attentions,)
        return outputs


class BertDecoderDisentangled(nn.Module):
    def __init__(self, config):
        super().__init__()
        if isinstance(config, dict):
            config = dict2obj(config)
        self.bert = BertDecoder(config)

    def get_word_embeddings(self):
        return self.bert.get_word_embeddings()

    def set_word_embeddings(self, we):
        self.bert.set_word_embeddings(we)

    def nacf_groups(self):
        return self.bert.nacf_groups()

    def nacf_bind(self, flat, rt):
        self.bert.nacf_bind(flat, rt)

    def project_memory(self, enc_output):
        return self.bert.project_memory(enc_output)

    def forward_(self, tgt_seq, enc_output, category, **kwargs):
        seq_probs, embs, *_ = self.bert(tgt_seq, enc_output, category, **kwargs)
        seq_probs = seq_probs[0]
        if len(_):
            return seq_probs, embs, _
        return seq_probs, embs

    def forward(self, tgt_seq, enc_output, category, **kwargs):
        if isinstance(enc_output, list):
            assert len(enc_output) == 1
            enc_output = enc_output[0]
        if isinstance(tgt_seq, (list, tuple)):
            # visual-word pass + caption pass: one launch sequence over 2B rows
            assert len(tgt_seq) == 2
            B = tgt_seq[0].shape[0]
            both = ops.stacked_rows([tgt_seq[0], tgt_seq[1]])       # a view when the two passes are adjacent in memory
            kwargs = dict(kwargs, row_map=('mod', enc_output.shape[0]))
            hidden, embs = self.forward_(both, enc_output, category, **kwargs)[:2]
            h0, h1 = HalvesFn.apply(hidden)
            return (BatchedPasses([h0, h1], hidden), None if embs is None else embs[B:],)
        return self.forward_(tgt_seq, enc_output, category, **kwargs)
