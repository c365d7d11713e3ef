"""NA decoding algorithms on the device (reference: decoding/algorithms.py).

`MaskPredict` (with / without coarse-grained templates), `Left2Right` and
`EasyFirst` keep the reference's class names and option keys, but a pass is
    decoder (cached memory K|V, no enlarge copies) -> fused vocab GEMM+softmax-max
and all per-row bookkeeping (select_worst top-k, re-masking, "only update masked
slots", PAD/probability fix-ups) runs in small index kernels, so an iteration
issues no host sync and no Python per-row loop
(reference: algorithms.py:206-215 loops over B*lbs rows on the host).
"""
import torch

from ..config import Constants
from ..runtime import ops


class DecodeContext(object):
    """What one NA pass needs: model, memory, its cached projections, row map."""

    def __init__(self, model, enc_output, category, pooled, lbs):
        self.model = model
        self.enc_output = enc_output
        self.category = category
        self.pooled = pooled
        self.lbs = lbs
        self.memory_kv = model.decoder.project_memory(enc_output)
        self.vocab_w = model.tgt_word_prj.weight
        self.vocab_b = model.tgt_word_prj.bias
        self.row_set = None     # live slots of the canvas (set by the algorithm once the canvas exists)

    def hidden(self, tokens, decoding_type='NARFormer', output_attentions=False, row_set='canvas', out_row_set=None):
        out = self.model.decoder(tokens, enc_output=self.enc_output, category=self.category,
                                 decoding_type=decoding_type, row_map=('div', self.lbs),
                                 memory_kv=self.memory_kv, pooled_memory=self.pooled,
                                 output_attentions=output_attentions,
                                 row_set=self.row_set if row_set == 'canvas' else row_set, out_row_set=out_row_set)
        h = out[0]
        if isinstance(h, list):
            h = h[-1]
        return h, (out[2] if output_attentions else None)


_LUT_CACHE = {}      # (ratios, width, device) -> device table; module level so that a captured decode never uploads


class Algorithm_Base(object):
    def __init__(self, opt, dict_mapping, tgt_vocab):
        self.opt = opt
        self.collect_best_candidate_iterative_results = opt.get('collect_best_candidate_iterative_results', False)
        self.collect_last = opt.get('collect_last', False)
        self.collect_results, self.collect_scores = [], []
        self.dict_mapping = dict_mapping
        self.masking_decision = opt.get('masking_decision', False)
        self.no_candidate_decision = opt.get('no_candidate_decision', False)
        self.vocab = tgt_vocab
        # [open slots before the last pass, open slots after it] of a fixed-pass-count run (EasyFirst; see static_passes)
        self.status = None

    def static_passes(self):
        """True while the decode is being captured into a hipGraph (or opt['decode_fixed_passes'], for tests): the
        data-dependent pass counts of Left2Right / EasyFirst are replaced by their upper bounds -- the extra passes
        select no slot and change nothing -- so that the launch sequence is fixed and no host read sits between passes"""
        return bool(self.opt.get('decode_fixed_passes', False)) or torch.cuda.is_current_stream_capturing()

    # ------------------------------------------------------------ helpers
    def collect_data(self, tokens, probs, is_last=False):
        if self.collect_best_candidate_iterative_results and not self.collect_last:
            self.collect_results.append(tokens.clone())
            self.collect_scores.append(probs.clone())
        elif self.collect_last and is_last:
            self.collect_results.append(tokens.clone())
            self.collect_scores.append(probs.clone())

    def get_collected_data(self):
        return self.collect_results, self.collect_scores, [[], []]

    def na_pass(self, ctx, tokens, probs, pad_tokens, update_mask=None, zero_mask_prob=False,
                out_tokens=None, out_probs=None):
        """generate_non_autoregressive (algorithms.py:143-167): decoder pass + fused
        projection/softmax/max; results land in (out_tokens, out_probs) or in place."""
        if ctx.row_set is None:
            # canvas slots that are not <pad>: fixed for the whole decode (pad_mask never changes), so the
            # decoder GEMMs of every pass skip the 35-60 % of slots beyond each candidate's length
            ctx.row_set = ops.rowset_build(tokens=pad_tokens.reshape(-1))
            if out_probs is None:
                ops.init_probs(pad_tokens.reshape(-1), probs.reshape(-1))   # <pad> slots: (PAD, 1.0) for good
        # only the slots this pass may change are read from the decoder and projected: live canvas slots, and within
        # them the re-masked ones (keys / values of the self-attention still cover every live slot)
        live = ctx.row_set if update_mask is None else ops.rowset_build(tokens=pad_tokens.reshape(-1),
                                                                        flags=update_mask.reshape(-1))
        h, _ = ctx.hidden(tokens, out_row_set=None if update_mask is None else live)
        R, Lp, D = h.shape
        ops.vocab_argmax(h.reshape(R * Lp, D), ctx.vocab_w, ctx.vocab_b, pad_tokens.reshape(-1), zero_mask_prob,
                         update_mask.reshape(-1) if update_mask is not None else None,
                         (out_tokens if out_tokens is not None else tokens).reshape(-1),
                         (out_probs if out_probs is not None else probs).reshape(-1), rows=live)

    def num_mask_lut(self, ratios, Lp, device):
        """floor(len * ratio) exactly as `(seq_lens.float() * ratio).long()` computes it
        (algorithms.py:256), tabulated per length so the kernel needs no float math."""
        key = (tuple(ratios), Lp, str(device))
        if key not in _LUT_CACHE:
            lens = torch.arange(Lp + 1, dtype=torch.float32)
            rows = [(lens * r).long() for r in ratios]
            _LUT_CACHE[key] = torch.stack(rows, 0).to(torch.int32).to(device)
        return _LUT_CACHE[key]

    def scoring_by_teacher(self, teacher_ctx, tokens, pad_tokens, is_last=False):
        """algorithms.py:169-204: p(y_t | y_<t) under an autoregressive teacher."""
        if teacher_ctx is None:
            return None
        if is_last and self.no_candidate_decision:
            return None
        if (not is_last) and not self.masking_decision:
            return None
        toks = tokens
        if self.dict_mapping:
            lut = torch.zeros(max(self.dict_mapping.keys()) + 1, dtype=torch.int64)
            for k, v in self.dict_mapping.items():
                lut[k] = v
            toks = lut.to(tokens.device)[tokens]
        R, Lp = toks.shape
        with_bos = torch.cat([toks.new_full((R, 1), Constants.BOS), toks[:, :-1]], dim=1).contiguous()
        h, _ = teacher_ctx.hidden(with_bos, decoding_type='ARFormer', row_set=None)
        V = teacher_ctx.vocab_w.shape[0]
        buf = torch.empty(R * Lp, ops.vocab_ld(V), device=h.device)
        logits = buf[:, :V]
        ops.linear_fwd(h.reshape(R * Lp, -1), teacher_ctx.vocab_w, logits, ops.Epi(bias=teacher_ctx.vocab_b))
        label_logp = torch.empty(R * Lp, device=h.device)
        ops.vocab_logsoftmax_fwd(logits, V, toks.reshape(-1).contiguous(), None, None, label_logp)
        out = torch.empty(R, Lp, device=h.device)
        ops.teacher_probs(label_logp, pad_tokens.reshape(-1), out.reshape(-1))
        return out


class MaskPredict(Algorithm_Base):
    """algorithms.py:224-273"""

    def __init__(self, opt, dict_mapping, tgt_vocab):
        super().__init__(opt, dict_mapping, tgt_vocab)
        self.use_ct = opt.get('use_ct', False)
        self.T = opt.get('iterations', 5)

    def generate(self, ctx, teacher_ctx, tokens):
        R, Lp = tokens.shape
        dev = tokens.device
        pad_tokens = tokens.clone()                      # the initial canvas fixes pad_mask / seq_lens
        probs = torch.empty(R, Lp, dtype=torch.float32, device=dev)
        mask = torch.empty(R, Lp, dtype=torch.uint8, device=dev)
        T = self.T + 1 if self.use_ct else self.T
        if self.use_ct:                                  # coarse-grained templates (algorithms.py:136-141)
            ops.token_replace(tokens, Constants.MASK, Constants.VIS)
            self.na_pass(ctx, tokens, probs, pad_tokens, zero_mask_prob=True)
        else:
            self.na_pass(ctx, tokens, probs, pad_tokens)
        self.collect_data(tokens, probs)
        lut = self.num_mask_lut([1.0 - (c / T) for c in range(T)], Lp, dev)
        for counter in range(1, T):
            tp = self.scoring_by_teacher(teacher_ctx, tokens, pad_tokens, is_last=False)
            if self.use_ct and counter == 1:
                ops.select_mask(None, None, pad_tokens, None, 1, tokens, mask)   # mask = tokens == MASK
            else:
                ops.select_mask(probs, tp, pad_tokens, lut[counter], 0, tokens, mask)
            self.na_pass(ctx, tokens, probs, pad_tokens, update_mask=mask)
            self.collect_data(tokens, probs, is_last=(counter == T - 1))
        tp = self.scoring_by_teacher(teacher_ctx, tokens, pad_tokens, is_last=True)
        return tokens, probs, tp, self.get_collected_data()


class _QFill(Algorithm_Base):
    """shared machinery of Left2Right / EasyFirst (algorithms.py:275-418)"""

    def __init__(self, opt, dict_mapping, tgt_vocab):
        super().__init__(opt, dict_mapping, tgt_vocab)
        self.use_ct = opt.get('use_ct', False)
        self.T = opt.get('q_iterations', 1)
        self.q = opt.get('q', 1)

    def _start(self, ctx, tokens):
        R, Lp = tokens.shape
        dev = tokens.device
        pad_tokens = tokens.clone()
        probs = torch.empty(R, Lp, dtype=torch.float32, device=dev)
        mask = torch.empty(R, Lp, dtype=torch.uint8, device=dev)
        visual_mask = None
        if self.use_ct:
            ops.token_replace(tokens, Constants.MASK, Constants.VIS)
            self.na_pass(ctx, tokens, probs, pad_tokens, zero_mask_prob=True)
            visual_mask = torch.empty(R, Lp, dtype=torch.uint8, device=dev)
            scratch = tokens.clone()
            ops.select_mask(None, None, pad_tokens, None, 2, scratch, visual_mask)  # != MASK and != PAD
        else:
            ops.init_probs(pad_tokens, probs)            # 0 everywhere, 1 on PAD (algorithms.py:294-295)
        self.collect_data(tokens, probs)
        return pad_tokens, probs, mask, visual_mask

    def _refine(self, ctx, tokens, probs, pad_tokens, mask, visual_mask):
        Lp = tokens.shape[1]
        lut = self.num_mask_lut([0.4 * (1.0 - (i / self.T)) for i in range(self.T)], Lp, tokens.device)
        for i in range(self.T):
            if i == 0 and self.use_ct:
                ops.apply_mask(tokens, visual_mask, Constants.MASK)
                m = visual_mask
            else:
                ops.select_mask(probs, None, pad_tokens, lut[i], 0, tokens, mask)
                m = mask
            self.na_pass(ctx, tokens, probs, pad_tokens, update_mask=m)
            self.collect_data(tokens, probs)


class Left2Right(_QFill):
    def generate(self, ctx, teacher_ctx, tokens):
        pad_tokens, probs, mask, visual_mask = self._start(ctx, tokens)
        R, Lp = tokens.shape
        rank = torch.empty(R, Lp, dtype=torch.int32, device=tokens.device)
        counts = torch.empty(2, dtype=torch.int32, device=tokens.device)
        ops.mask_rank(tokens, rank, counts)              # rank of every <mask> slot within its row
        # one host read replaces the per-step `mask_ind.sum() == 0`; without it (graph capture) every rank up to the canvas
        # width is visited: a pass past a row's last <mask> selects nothing in that row
        n_slots = Lp if self.static_passes() else int(counts[0].item())
        for cur in range(0, min(n_slots, Lp), self.q):
            ops.select_rank(rank, cur, self.q, tokens, mask)
            self.na_pass(ctx, tokens, probs, pad_tokens, update_mask=mask)
            self.collect_data(tokens, probs)
        self._refine(ctx, tokens, probs, pad_tokens, mask, visual_mask)
        tp = self.scoring_by_teacher(teacher_ctx, tokens, pad_tokens, is_last=True)
        return tokens, probs, tp, self.get_collected_data()


class EasyFirst(_QFill):
    def generate(self, ctx, teacher_ctx, tokens):
        pad_tokens, probs, mask, visual_mask = self._start(ctx, tokens)
        R, Lp = tokens.shape
        rank = torch.empty(R, Lp, dtype=torch.int32, device=tokens.device)
        counts = torch.empty(2, dtype=torch.int32, device=tokens.device)
        new_tokens = torch.empty_like(tokens)
        new_probs = torch.empty_like(probs)
        pre = 0
        if self.static_passes():
            # fixed launch sequence: ceil(Lp / q) passes fill every row that makes progress; further passes on a finished or
            # stalled canvas recompute what is there.  A slot may also stay open because its prediction WAS the <mask> id
            # (then the reference simply runs more passes): the open-slot counts before and after the last pass go to
            # `self.status`, and the caller falls back to the host-driven loop when they show an unfinished, moving canvas
            n_pass = (Lp + self.q - 1) // self.q
            self.status = torch.zeros(2, dtype=torch.int32, device=tokens.device)
            for i in range(n_pass):
                if i == n_pass - 1:
                    ops.mask_rank(tokens, rank, counts)
                    self.status[0:1].copy_(counts[1:2])
                self.na_pass(ctx, tokens, probs, pad_tokens, out_tokens=new_tokens, out_probs=new_probs)
                ops.easy_first_update(tokens, probs, new_tokens, new_probs, self.q)
                self.collect_data(tokens, probs)
            ops.mask_rank(tokens, rank, counts)
            self.status[1:2].copy_(counts[1:2])
            self._refine(ctx, tokens, probs, pad_tokens, mask, visual_mask)
            tp = self.scoring_by_teacher(teacher_ctx, tokens, pad_tokens, is_last=True)
            return tokens, probs, tp, self.get_collected_data()
        while True:
            # the reference reads `mask_ind.sum()` on the host every pass as well (algorithms.py:380-385):
            # a slot whose prediction is the <mask> id itself stays open, so the pass count is data dependent
            ops.mask_rank(tokens, rank, counts)
            remain = int(counts[1].item())
            if remain == 0 or pre == remain:
                break
            pre = remain
            self.na_pass(ctx, tokens, probs, pad_tokens, out_tokens=new_tokens, out_probs=new_probs)
            ops.easy_first_update(tokens, probs, new_tokens, new_probs, self.q)
            self.collect_data(tokens, probs)
        self._refine(ctx, tokens, probs, pad_tokens, mask, visual_mask)
        tp = self.scoring_by_teacher(teacher_ctx, tokens, pad_tokens, is_last=True)
        return tokens, probs, tp, self.get_collected_data()
