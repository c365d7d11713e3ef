"""Auxiliary length predictor (reference: models/Predictor.py:12-43):
mean_t(enc_output) -> Linear -> ReLU -> Dropout -> Linear -> log_softmax,
run as two small GEMMs with fused epilogues + a row log-softmax kernel."""
import torch.nn as nn

from ..runtime.functional import LengthHeadFn, MeanTimeFn

__all__ = ('Predictor_length', 'Auxiliary_Task_Predictor')


class Predictor_length(nn.Module):
    def __init__(self, opt, key_name):
        super().__init__()
        self.p = opt['hidden_dropout_prob']
        self.net = nn.Sequential(nn.Linear(opt['dim_hidden'], opt['dim_hidden']), nn.ReLU(), nn.Dropout(self.p),
                                 nn.Linear(opt['dim_hidden'], opt['max_len']))
        self.key_name = key_name

    def nacf_groups(self):
        return [[self.net[0].weight], [self.net[0].bias], [self.net[3].weight], [self.net[3].bias]]

    def nacf_bind(self, flat, rt):
        self._rt = rt
        self._cfg = dict(l1=flat.pack([self.net[0].weight], [self.net[0].bias], image='both'),
                         l2=flat.pack([self.net[3].weight], [self.net[3].bias], image='both'), p=self.p, salt=rt.next_salt())

    def forward(self, enc_output, pooled=None, **kwargs):
        if isinstance(enc_output, list):
            assert len(enc_output) == 1
            enc_output = enc_output[0]
        assert enc_output.dim() == 3
        if pooled is None:
            pooled = MeanTimeFn.apply(enc_output)
        cfg = dict(self._cfg, training=self.training, rng=self._rt.rng(enc_output.device))
        params = [self.net[0].weight, self.net[0].bias, self.net[3].weight, self.net[3].bias]
        return {self.key_name: LengthHeadFn.apply(pooled, cfg, *params)}


class Auxiliary_Task_Predictor(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.layers = nn.ModuleList(layers)

    def nacf_groups(self):
        return [g for l in self.layers for g in l.nacf_groups()]

    def nacf_bind(self, flat, rt):
        for l in self.layers:
            l.nacf_bind(flat, rt)

    def forward(self, enc_output, **kwargs):
        results = {}
        for layer in self.layers:
            results.update(layer(enc_output=enc_output, **kwargs))
        return results
