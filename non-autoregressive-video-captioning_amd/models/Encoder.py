"""Visual encoder: per modality Linear -> HighWay -> Dropout.

Drop-in for the reference's `models.Encoder` registry (models/Encoder.py:5-7,
62-66): same class name, same sub-module / parameter names
(`Encoder_M.0.weight`, `Encoder_M.1.w1.weight`, ...), same construction order
(so torch's default initialisers consume the RNG identically).  The modules
below only HOLD parameters; the compute is `EncoderStreamFn` (two MFMA GEMMs
with fused bias / tanh|sigmoid epilogues + the gate-mix/dropout kernel), with
HighWay's w1|w2 packed into one [2D, D] GEMM.
"""
import torch.nn as nn

from ..runtime import lib as L
from ..runtime.functional import EncoderStreamFn, EncoderStreamsFn, LinearFn

__all__ = ('Encoder_HighWay',)


class HighWay(nn.Module):
    """parameter holder for the gated HighWay layer (models/Encoder.py:9-25)"""

    def __init__(self, hidden_size, with_gate=True):
        super().__init__()
        self.with_gate = with_gate
        self.w1 = nn.Linear(hidden_size, hidden_size)
        if with_gate:
            self.w2 = nn.Linear(hidden_size, hidden_size)


class Encoder_HighWay(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.modality = opt['modality'].lower()
        self.dropout = opt.get('encoder_dropout', 0.5)
        dim_hidden = opt.get('dim_hidden', 512)
        self.streams = []
        for char in self.modality:
            input_dim = opt.get('dim_' + char, None)
            assert input_dim is not None, 'modality %s needs dim_%s in opt' % (self.modality, char)
            seq = nn.Sequential(nn.Linear(input_dim, dim_hidden), HighWay(dim_hidden, opt.get('gate', True)),
                                nn.Dropout(self.dropout))
            self.add_module('Encoder_%s' % char.upper(), seq)
            self.streams.append(seq)
        self.num_feats = len(self.modality)
        self.joint_streams = bool(opt.get('encoder_joint_streams', True))     # nacf_amd option: False = one autograd node per modality
        self._cfg = None

    # --- flat-buffer binding -------------------------------------------------
    def nacf_groups(self):
        groups = []
        for s in self.streams:
            if s[1].with_gate:
                groups += [[s[0].weight], [s[0].bias], [s[1].w1.weight, s[1].w2.weight], [s[1].w1.bias, s[1].w2.bias]]
            else:
                groups += [[s[0].weight], [s[0].bias], [s[1].w1.weight], [s[1].w1.bias]]
        return groups

    def nacf_bind(self, flat, rt):
        self._rt = rt
        self._cfg = []
        for s in self.streams:
            if not s[1].with_gate:
                # opt['gate'] = False (models/Encoder.py:24-25): out = dropout(h + tanh(w1 h)) is ONE Linear with the fused
                # epilogue tanh -> + residual -> dropout (nacf_linear_fwd); two LinearFn nodes per stream
                self._cfg.append(dict(gate=False, lin=dict(pack=flat.pack([s[0].weight], [s[0].bias], image='fwd')),
                                      hw=dict(pack=flat.pack([s[1].w1.weight], [s[1].w1.bias], image='both'), act=L.ACT_TANH,
                                              p2=self.dropout, salt2=rt.next_salt())))
                continue
            self._cfg.append(dict(lin=flat.pack([s[0].weight], [s[0].bias], image='fwd'),    # no dX: the features need no gradient
                                  hw=flat.pack([s[1].w1.weight, s[1].w2.weight], [s[1].w1.bias, s[1].w2.bias], image='both'),
                                  p=self.dropout, salt=rt.next_salt(),
                                  params=[s[0].weight, s[0].bias, s[1].w1.weight, s[1].w1.bias, s[1].w2.weight,
                                          s[1].w2.bias]))

    def forward(self, input_feats):
        assert self.num_feats == len(input_feats)
        if not self._cfg[0].get('gate', True):
            outs = []
            for cfg, x, st in zip(self._cfg, input_feats, self.streams):
                B, F, Din = x.shape
                rng = self._rt.rng(x.device)
                h = LinearFn.apply(x.reshape(B * F, Din), None, dict(cfg['lin'], training=self.training, rng=rng),
                                   st[0].weight, st[0].bias)
                o = LinearFn.apply(h, h, dict(cfg['hw'], training=self.training, rng=rng), st[1].w1.weight, st[1].w1.bias)
                outs.append(o.view(B, F, -1))
            return outs, None
        cs = [dict(cfg, training=self.training, rng=self._rt.rng(x.device)) for cfg, x in zip(self._cfg, input_feats)]
        if self.num_feats > 1 and self.joint_streams:
            # layer by layer across the modalities: their independent GEMMs share launches (EncoderStreamsFn)
            params = [p for cfg in self._cfg for p in cfg['params']]
            outs = list(EncoderStreamsFn.apply(cs, self.num_feats, *input_feats, *params))
        else:
            outs = [EncoderStreamFn.apply(x, c, *cfg['params']) for cfg, c, x in zip(self._cfg, cs, input_feats)]
        return outs, None  # hiddens (mean over time) are produced lazily by Seq2Seq.encode
