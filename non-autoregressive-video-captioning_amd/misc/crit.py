"""Criterion for the training step (reference: misc/crit.py:10-251).

Same semantics: language loss = sum_i w_i * (token-SUM NLL of pass i) / B with
weights `nv_weights` when visual_word_generation (crit.py:40-55,82); length
loss = legacy nn.KLDivLoss() mean over B*max_len elements (crit.py:223); total
= sum_i scale_i * loss_i.  Side meters: top-1 word accuracy per pass (pass 0
excludes <mask> labels) and perplexity of the caption pass (crit.py:86-114).

Two input forms:
  * results['_nacf_hidden'] (model built with opt['fused_loss']=True): the
    vocabulary projection, log-softmax, NLL, argmax-hit and gathered-logp
    reductions run as ONE fused operator per pass; [B, L, V] log-probs are
    never handed to Python.
  * results['tgt_word_logprobs'] (reference form): consumed as tensors.
Meters accumulate on the device; the host only syncs in get_loss_info().
"""
import math
import os

import torch

from ..config import Constants
from ..runtime import ops
from ..runtime.functional import FusedVocabXentFn, FusedVocabXentMultiFn, KLDivMeanFn, LossCombineFn


class Criterion(object):
    def __init__(self, opt):
        self.opt = opt
        self.crit = [c.lower() for c in opt['crit']]
        default_names = {'lang': 'Cap Loss', 'length': 'Length Loss'}
        self.names = list(opt.get('crit_name', []))
        self.scales = list(opt.get('crit_scale', []))
        if len(self.names) != len(self.crit):      # hand-written opt dicts: opts.py:185-189 defaults
            self.names = [default_names[c] for c in self.crit]
        if len(self.scales) != len(self.crit):
            self.scales = [1.0] * len(self.crit)
        self.vw = opt.get('visual_word_generation', False)
        self.weights = list(opt.get('nv_weights', [0.8, 1.0])) if self.vw else None
        self.reset_loss_recorder()

    def reset_loss_recorder(self):
        self._loss_sum = [None] * len(self.crit)
        self._loss_cnt = [0] * len(self.crit)
        self._acc = None      # device [n_pass, 2] (hits, count)
        self._ppl = None      # device [2] (sum logp, count)
        if getattr(self, '_meters', None) is not None:
            self._meters.zero_()      # in place: a captured hipGraph keeps writing this very buffer
        else:
            self._meters = None       # fused form: device [n_crit | 2*n_pass | 2] running sums, see _fused_plan
            self._meter_plan = None
        if not hasattr(self, '_plans'):
            self._plans = {}

    @staticmethod
    def _acc_add(cur, new):
        return new if cur is None else cur + new

    def _lang(self, results, labels):
        if '_nacf_hidden' in results:
            hidden = results['_nacf_hidden']
            pack, params = results['_nacf_vocab']
        else:
            hidden = None
            logps = results[Constants.mapping['lang'][0]]
            if not isinstance(logps, (list, tuple)):
                logps = [logps]
        n = len(hidden) if hidden is not None else len(logps)
        if not isinstance(labels, (list, tuple)):
            labels = [labels] * n
        weights = self.weights if self.weights is not None else [1.0] * n
        assert len(labels) == n == len(weights)
        B = (hidden[0] if hidden is not None else logps[0]).shape[0]
        loss = None
        acc_rows, ppl = [], None
        for i in range(n):
            exclude = (i == 0 and self.vw)
            lab = labels[i].contiguous()
            if hidden is not None:
                h = hidden[i]
                assert h.shape[1] == lab.shape[1]
                stats = FusedVocabXentFn.apply(h.reshape(-1, h.shape[-1]), dict(pack=pack), lab, exclude, *params)
            else:
                lp = logps[i]
                assert lp.shape[1] == lab.shape[1]
                m = lab.ne(Constants.PAD)
                g = lp.gather(2, lab.unsqueeze(2)).squeeze(2)
                ind = m & lab.ne(Constants.MASK) if exclude else m
                hit = (lp.argmax(-1).eq(lab) & ind).sum().float()
                stats = torch.stack([-(g * m).sum(), hit, ind.sum().float(), (g * m).sum().detach(), m.sum().float()])
            term = weights[i] * stats[0] / B
            loss = term if loss is None else loss + term
            acc_rows.append(stats[1:3].detach())
            if not exclude:
                ppl = stats[3:5].detach()
        self._acc = self._acc_add(self._acc, torch.stack(acc_rows))
        if ppl is not None:
            self._ppl = self._acc_add(self._ppl, ppl)
        return loss, B

    # ---- fused form: every term lands in one device slab, ONE kernel forms the total and updates the meters
    _STRIDE = 8

    def _fused_plan(self, B, n_pass, n_len, device):
        key = (B, n_pass, n_len, str(device))
        plan = self._plans.get(key)
        if plan is not None:
            return plan
        weights = self.weights if self.weights is not None else [1.0] * n_pass
        assert len(weights) == n_pass
        n_crit = len(self.crit)
        acc_base = n_crit
        ppl_base = n_crit + 2 * n_pass
        coef, m_dst, m_src, m_scale, slots = [], [], [], [], []
        t = 0
        for ci, name in enumerate(self.crit):
            if name == 'lang':
                for i in range(n_pass):
                    coef.append(self.scales[ci] * weights[i] / B)
                    m_dst.append(ci); m_src.append(t * self._STRIDE); m_scale.append(weights[i])     # loss sum (li * B)
                    m_dst += [acc_base + 2 * i, acc_base + 2 * i + 1]
                    m_src += [t * self._STRIDE + 1, t * self._STRIDE + 2]; m_scale += [1.0, 1.0]
                    if i == n_pass - 1:        # perplexity: the last pass that does not exclude <mask> labels
                        m_dst += [ppl_base, ppl_base + 1]
                        m_src += [t * self._STRIDE + 3, t * self._STRIDE + 4]; m_scale += [1.0, 1.0]
                    slots.append(('lang', i, t))
                    t += 1
            elif name == 'length':
                coef.append(self.scales[ci])
                m_dst.append(ci); m_src.append(t * self._STRIDE); m_scale.append(float(n_len))
                slots.append(('length', 0, t))
                t += 1
            else:
                raise NotImplementedError('criterion %s' % name)
        f = lambda v, dt: torch.tensor(v, dtype=dt, device=device)
        plan = dict(n_terms=t, slots=slots, coef=f(coef, torch.float32), m_dst=f(m_dst, torch.int32),
                    m_src=f(m_src, torch.int32), m_scale=f(m_scale, torch.float32), n_pass=n_pass,
                    acc_base=acc_base, ppl_base=ppl_base)
        self._plans[key] = plan
        return plan

    def _get_loss_fused(self, results):
        hidden = results['_nacf_hidden']
        pack, params = results['_nacf_vocab']
        labels = results[Constants.mapping['lang'][1]]
        n_pass = len(hidden)
        if not isinstance(labels, (list, tuple)):
            labels = [labels] * n_pass
        dev = hidden[0].device
        B = hidden[0].shape[0]
        pred = results.get(Constants.mapping['length'][0]) if 'length' in self.crit else None
        plan = self._fused_plan(B, n_pass, pred.shape[0] if pred is not None else 0, dev)
        if self._meters is None:
            self._meters = torch.zeros(plan['ppl_base'] + 2, dtype=torch.float32, device=dev)
            self._meter_plan = plan
        S = self._STRIDE
        slab = torch.empty(plan['n_terms'] * S, dtype=torch.float32, device=dev)
        terms = []
        # the terms' producers leave their reductions to ONE tail launch (nacf_crit_tail_fwd: per-pass criterion scalars + the
        # length head's KL + the weighted total and the meters; backward: one launch too) -- NACF_CRIT_TAIL=0: a launch each
        tail = ops.CritTail() if os.environ.get('NACF_CRIT_TAIL', '1') != '0' else None
        both = getattr(hidden, 'both', None)      # the passes are halves of one tensor: project them in one go
        lang_terms = None
        if both is not None and all(labels[i].shape == labels[0].shape for i in range(n_pass)):
            lang_slots = [t for kind, _, t in plan['slots'] if kind == 'lang']
            lab_all = ops.stacked_rows([labels[i].contiguous() for i in range(n_pass)]).reshape(-1)
            lang_terms = FusedVocabXentMultiFn.apply(
                both.reshape(-1, both.shape[-1]), dict(pack=pack, outs=[slab[t * S:t * S + 5] for t in lang_slots], tail=tail, slots=lang_slots),
                lab_all, tuple((i == 0 and self.vw) for i in range(n_pass)), *params)
        for kind, i, t in plan['slots']:
            if kind == 'lang' and lang_terms is not None:
                terms.append(lang_terms[i])
            elif kind == 'lang':
                h, lab = hidden[i], labels[i].contiguous()
                assert h.shape[1] == lab.shape[1]
                terms.append(FusedVocabXentFn.apply(h.reshape(-1, h.shape[-1]), dict(pack=pack, out=slab[t * S:t * S + 5], tail=tail, slot=t),
                                                    lab, (i == 0 and self.vw), *params))
            else:
                terms.append(KLDivMeanFn.apply(pred, results[Constants.mapping['length'][1]].to(pred.dtype),
                                               slab[t * S:t * S + 1], tail, t))
        for ci, name in enumerate(self.crit):
            self._loss_cnt[ci] += B if name == 'lang' else pred.shape[0]
        cfg = dict(slab=slab, stride=S, coef=plan['coef'], m_dst=plan['m_dst'], m_src=plan['m_src'],
                   m_scale=plan['m_scale'], meters=self._meters, tail=tail)
        return LossCombineFn.apply(cfg, *terms)

    def get_loss(self, results, **kwargs):
        if '_nacf_hidden' in results and 'lang' in self.crit:
            return self._get_loss_fused(results)
        total = None
        for i, name in enumerate(self.crit):
            if name == 'lang':
                li, n = self._lang(results, results[Constants.mapping['lang'][1]])
            elif name == 'length':
                pred = results[Constants.mapping['length'][0]]
                li = KLDivMeanFn.apply(pred, results[Constants.mapping['length'][1]].to(pred.dtype))
                n = pred.shape[0]
            else:
                raise NotImplementedError('criterion %s' % name)
            self._loss_sum[i] = self._acc_add(self._loss_sum[i], li.detach() * n)
            self._loss_cnt[i] += n
            term = li * self.scales[i]
            total = term if total is None else total + term
        return total

    def get_fieldsnames(self):
        """CSV columns this criterion adds to trainning_record.csv: the loss names of every term that is NOT the
        language-generation one, as the reference (misc/crit.py:199-209) -- e.g. ['Length Loss']"""
        return [n for n, c in zip(self.names, self.crit) if c != 'lang']

    def get_loss_info(self):
        names = list(self.names)
        sums = [float(s) if s is not None else 0.0 for s in self._loss_sum]
        acc = self._acc.tolist() if self._acc is not None else None
        ppl = self._ppl.tolist() if self._ppl is not None else None
        if self._meters is not None:         # fused form: one read-back of the meter vector
            m, pl = self._meters.tolist(), self._meter_plan
            for ci in range(len(self.crit)):
                sums[ci] += m[ci]
            a = [[m[pl['acc_base'] + 2 * i], m[pl['acc_base'] + 2 * i + 1]] for i in range(pl['n_pass'])]
            acc = a if acc is None else [[x[0] + y[0], x[1] + y[1]] for x, y in zip(acc, a)]
            p = [m[pl['ppl_base']], m[pl['ppl_base'] + 1]]
            ppl = p if ppl is None else [ppl[0] + p[0], ppl[1] + p[1]]
        info = [s / max(c, 1) for s, c in zip(sums, self._loss_cnt)]
        if acc is not None:
            for i, (h, c) in enumerate(acc):
                names.append('Word Acc%d' % i)
                info.append(h / c if c > 0 else 0.0)   # the reference divides by zero here
        if ppl is not None:
            s, c = ppl
            names.append('Perplexity')
            info.append(math.exp(-s / c) if c > 0 else float('nan'))
        return names, info


def get_criterion(opt, summarywriter=None):
    assert isinstance(opt['crit'], list)
    return Criterion(opt)
