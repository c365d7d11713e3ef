"""Seq2Seq wrapper: same surface as the reference (models/seq2seq.py:7-140) --
`encode`, `prepare_inputs_for_decoder`, `forward(**kwargs)` dispatching on
opt['decoding_type'] -- over the HIP operators.

Additions that do not change the reference call convention:
  * every parameter lives in ONE flat fp32 buffer (`self.flat.data`) and every
    gradient in a second one (`self.flat.grad`): a single RCCL all-reduce
    bucket and a single fused clip+Adam launch;
  * opt['fused_loss'] (default False): skip materialising the [B, L, V]
    log-probs and hand the criterion (`nacf_amd.misc.crit`) the hidden states,
    which it turns into loss / word-accuracy / perplexity with the fused
    vocabulary-projection + NLL operator.
"""
import torch
import torch.nn as nn

from ..config import Constants
from ..runtime import ops
from ..runtime.functional import MeanTimeFn, MemoryFanoutFn, VocabLogProbFn
from ..runtime.state import FlatParams, Runtime


class LazyResults(dict):
    """the result dict of Seq2Seq.encode / forward with values that are computed on first access (`d[k]`, `d.get(k)`,
    `k in d`); they are not listed by keys() / items() / a plain dict(d) copy until then"""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._lazy = {}

    def lazy(self, key, fn):
        self._lazy[key] = fn

    def __missing__(self, key):
        fn = self._lazy.pop(key, None)
        if fn is None:
            raise KeyError(key)
        self[key] = v = fn()
        return v

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._lazy


class Seq2Seq(nn.Module):
    def __init__(self, opt, preEncoder=None, encoder=None, joint_representation_learner=None,
                 auxiliary_task_predictor=None, decoder=None, tgt_word_prj=None, **kwargs):
        super().__init__()
        assert preEncoder is None
        self.opt = opt
        self.preEncoder = preEncoder
        self.encoder = encoder
        self.joint_representation_learner = joint_representation_learner
        self.auxiliary_task_predictor = auxiliary_task_predictor
        self.decoder = decoder
        self.tgt_word_prj = tgt_word_prj
        if opt.get('tie_weights', False):
            self._tie_weights(opt['vocab_size'])
        self.rt = Runtime(seed=opt.get('seed', 0))
        self.flat = None
        self._flatten()

    def _tie_weights(self, vocab_size):
        word_embeddings = self.decoder.get_word_embeddings()
        self.tgt_word_prj.weight = word_embeddings.weight
        self.tgt_word_prj.bias = nn.Parameter(torch.zeros(vocab_size).float(), requires_grad=True)

    # ------------------------------------------------------------- flat buffers
    def _parts(self):
        return [m for m in (self.encoder, self.joint_representation_learner, self.auxiliary_task_predictor,
                            self.decoder) if m is not None]

    def _flatten(self):
        if self.flat is not None and self.flat.images is not None:
            self.flat.images.close()
        groups = [g for m in self._parts() for g in m.nacf_groups()]
        groups.append([self.tgt_word_prj.weight])
        if self.tgt_word_prj.bias is not None:
            groups.append([self.tgt_word_prj.bias])
        covered = {id(p) for g in groups for p in g}
        missing = [n for n, p in self.named_parameters() if id(p) not in covered]
        assert not missing, 'parameters outside the flat layout: %s' % missing
        self.flat = FlatParams(groups)
        for m in self._parts():
            m.nacf_bind(self.flat, self.rt)
        self._vocab_pack = self.flat.pack([self.tgt_word_prj.weight],
                                          [self.tgt_word_prj.bias] if self.tgt_word_prj.bias is not None else None,
                                          image='both')

    def _apply(self, fn, *args, **kwargs):
        before = (self.flat.data.device, self.flat.data.dtype, self.flat.data.data_ptr())
        out = super()._apply(fn, *args, **kwargs)
        p0 = self.flat.params[0]
        moved = (p0.device, p0.dtype) != before[:2] or not self._views_intact()
        if moved:
            # .to(device) / .cuda() re-homed the tensors: rebuild the flat views there.  A no-op call (.to(same device),
            # .float()) must NOT re-flatten: captured step graphs and the optimiser's moments are tied to the buffers.
            self._flatten()
        return out

    def _views_intact(self):
        base = self.flat.data
        lo, hi = base.data_ptr(), base.data_ptr() + base.numel() * base.element_size()
        return all(lo <= p.data_ptr() < hi for p in self.flat.params)

    def head_parameters(self):
        """parameters between the decoder's output and the loss (the vocabulary projection) -- unless it is tied to the
        word embedding, whose gradient is only complete at the end of the decoder's backward"""
        if self.opt.get('tie_weights', False):
            return []
        return list(self.tgt_word_prj.parameters())

    _cut_head = None

    def late_parameters(self):
        """parameters whose gradients are complete once backward has reached the encoder output
        (length head + decoder + vocabulary projection): the first bucket of the overlapped all-reduce"""
        aux = list(self.auxiliary_task_predictor.parameters()) if self.auxiliary_task_predictor is not None else []
        return aux + list(self.decoder.parameters()) + list(self.tgt_word_prj.parameters())

    def zero_grad(self, set_to_none=False):
        self.flat.attach_grads(zero=True)

    def _ensure_grads(self):
        if not self.flat.grads_attached():
            self.flat.attach_grads(zero=True)

    # ------------------------------------------------------------- reference surface
    def encode(self, feats, **kwargs):
        results = LazyResults()
        if self.opt.get('automatic_mask', False):
            # models/seq2seq.py:37-42: results['attention_mask'] = [feat.sum(-1).eq(0) per modality].  Nothing in the reference reads
            # the entry (models/Decoder.py builds its masks from the tokens; the cross-attention takes none), so it costs nothing
            # here either: formed on first access, off the captured path
            frames = [f for f in feats]
            results.lazy('attention_mask', lambda: [f.detach().sum(-1).eq(0) for f in frames])
        # bf16 GEMM modes: the weight images follow the fp32 master weights at every training forward entry (one launch, part
        # of a captured step; the forward's and the backward's GEMMs then read images of exactly the weights an fp32 kernel
        # would read).  Inference rebuilds them only when the weights were written since (FlatParams.version: optimiser
        # step, step replay, load_state_dict, broadcast) -- 37 us per decode batch otherwise
        if self.training or torch.is_grad_enabled():
            self.flat.sync_images()
        else:
            self.flat.ensure_images()
        enc_streams, _ = self.encoder([f.contiguous() for f in feats])
        def enc_hidden_fn(streams=[s_.detach() for s_ in enc_streams]):
            # mean-over-time hidden of every stream, then the mean over the modalities (models/Encoder.py:51,
            # models/joint_representation.py:27).  Only RNN decoders consume it (seq2seq.py:66-68) and none is built here,
            # so it is computed on first ACCESS (`results['enc_hidden']`), not in every step: three launches less.
            with torch.no_grad():
                s0 = streams[0]
                hid = torch.empty(len(streams), s0.shape[0], s0.shape[2], dtype=s0.dtype, device=s0.device)
                for i, s_ in enumerate(streams):
                    ops.mean_time_fwd(s_.contiguous(), hid[i])
                return ops.mean_time_fwd(hid.view(1, hid.shape[0], -1), torch.empty_like(hid[0]).view(1, -1)).view_as(hid[0])
        enc_hidden = None
        if self.joint_representation_learner is not None:
            enc_output, enc_hidden = self.joint_representation_learner(enc_streams, enc_hidden)
        else:
            enc_output = torch.cat(enc_streams, dim=1)
        if self.training and torch.is_tensor(enc_output) and enc_output.requires_grad:
            results['_enc_root'] = enc_output          # (the staged backward cuts here: upstream of memory AND pooled)
            enc_output, pooled, pooled_dec = MemoryFanoutFn.apply(enc_output)     # (one mean, one handle per consumer)
        else:
            pooled = pooled_dec = MeanTimeFn.apply(enc_output)
        if self.auxiliary_task_predictor is not None:
            # (The length head -- ~130 us of launch-latency-sized kernels that depend on the pooled memory only -- was tried as a
            #  parallel branch of the step graph on a side stream: no difference, 2.765 vs 2.766 ms in one box; removed.)
            results.update(self.auxiliary_task_predictor(enc_output=enc_output, pooled=pooled))
        results['enc_output'] = enc_output
        results.lazy('enc_hidden', enc_hidden_fn)
        results['_pooled_memory'] = pooled_dec
        return results

    def prepare_inputs_for_decoder(self, encoder_outputs, category):
        inputs_for_decoder = {'category': category, 'enc_output': encoder_outputs['enc_output']}
        if isinstance(inputs_for_decoder['enc_output'], list):
            assert len(inputs_for_decoder['enc_output']) == 1
            inputs_for_decoder['enc_output'] = inputs_for_decoder['enc_output'][0]
        return inputs_for_decoder

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.flat.touch()
        return out

    def vocab_logprobs(self, hidden):
        """tgt_word_prj + log_softmax on [.., L, D] hidden states -> [.., L, V] log-probs"""
        self.flat.ensure_images()
        shape = hidden.shape
        params = [p for p in self.tgt_word_prj.parameters()]
        lp = VocabLogProbFn.apply(hidden.reshape(-1, shape[-1]), dict(pack=self._vocab_pack), *params)
        return lp.view(*shape[:-1], lp.shape[-1])

    def forward(self, **kwargs):
        func_name = 'forward_' + self.opt['decoding_type']
        fn = getattr(self, func_name, None)
        if fn is None:
            raise NotImplementedError('nacf_amd: decoding_type %s is not built' % self.opt['decoding_type'])
        return fn(kwargs)

    def _run(self, feats, tgt_tokens, category, decoding_type):
        if self.training:
            self._ensure_grads()
            self.flat.train_forwards += 1       # (host counter; a replayed captured step does not pass here)
            self.rt.rng(feats[0].device)
            self.rt.advance()
        results = self.encode(feats)
        if self.training:
            # the encoder/decoder boundary, for the staged backward of runtime/ddp.py: every path from the loss to an
            # encoder / fusion parameter goes through enc_output (the pooled memory and the length head hang off it),
            # so the gradients of everything downstream are complete before the encoder's backward starts
            eo = results.get('_enc_root', results['enc_output'])
            self._cut = [t for t in (eo if isinstance(eo, (list, tuple)) else [eo]) if t.requires_grad]
        inputs_for_decoder = self.prepare_inputs_for_decoder(results, category)
        hidden_states, embs, *_ = self.decoder(tgt_tokens, decoding_type=decoding_type, want_embs=False,
                                               pooled_memory=results['_pooled_memory'], **inputs_for_decoder)
        if not isinstance(hidden_states, list):
            hidden_states = [hidden_states]
        if self.opt.get('fused_loss', False):
            results['_nacf_hidden'] = hidden_states
            results['_nacf_vocab'] = (self._vocab_pack, [p for p in self.tgt_word_prj.parameters()])
            if self.training:
                # second boundary of the staged backward: everything between the loss and the decoder's output (the
                # vocabulary projection, 29 % of the gradient bytes) is complete once backward has reached these
                head = [getattr(hidden_states, 'both', None)]
                if head[0] is None:
                    head = list(hidden_states)
                pl = results.get(Constants.mapping['length'][0])
                self._cut_head = [t for t in head + ([pl] if torch.is_tensor(pl) else []) if t.requires_grad]
        else:
            results[Constants.mapping['lang'][0]] = [self.vocab_logprobs(h) for h in hidden_states]
        return results

    def forward_NARFormer(self, kwargs):
        feats, tgt_tokens, category = (kwargs.get(k, None) for k in ('feats', 'tgt_tokens', 'category'))
        return self._run(feats, tgt_tokens, category, 'NARFormer')

    def forward_ARFormer(self, kwargs):
        feats, tgt_tokens, category = (kwargs.get(k, None) for k in ('feats', 'tgt_tokens', 'category'))
        decoding_type = kwargs.get('decoding_type', self.opt['decoding_type'])
        if decoding_type != 'ARFormer':
            raise NotImplementedError('nacf_amd: decoding_type %s is not built' % decoding_type)
        if isinstance(tgt_tokens, (list, tuple)):   # teacher forcing: feed tokens[:, :-1] (seq2seq.py:120)
            tgt_tokens = [t[:, :-1].contiguous() for t in tgt_tokens]
        else:
            tgt_tokens = tgt_tokens[:, :-1].contiguous()
        return self._run(feats, tgt_tokens, category, 'ARFormer')
