"""Feature fusion: per-modality BatchNorm1d over the flattened B*F rows, then
temporal concatenation (reference: models/joint_representation.py:5-53).
One kernel sequence per modality normalises and writes straight into its slice
of the [B, sum F, D] memory (`BNConcatFn`)."""
import torch
import torch.nn as nn

from ..runtime.functional import BNConcatFn, LNConcatFn


class Joint_Representaion_Learner(nn.Module):  # (sic) upstream spelling is part of the state_dict contract
    def __init__(self, feats_size, opt):
        super().__init__()
        self.fusion = opt.get('fusion', 'temporal_concat')
        if self.fusion not in ('temporal_concat', 'none'):
            # 'addition' asserts inside the reference itself (joint_representation.py:41)
            raise ValueError('nacf_amd supports fusion temporal_concat | none (got %s)' % self.fusion)
        self.norm_list = []
        self.is_bn = opt.get('norm_type', 'bn').lower() == 'bn'
        if not opt['no_encoder_bn'] and self.fusion != 'none':
            for i, item in enumerate(feats_size):
                m = nn.BatchNorm1d(item) if self.is_bn else nn.LayerNorm(item)
                self.norm_list.append(m)
                self.add_module('%s%d' % ('bn' if self.is_bn else 'ln', i), m)
        self._packs = None
        self.sync_bn = bool(opt.get('sync_bn', False))
        self._sync = None            # set by runtime.ddp.DataParallel when opt['sync_bn'] and more than one rank trains

    def nacf_groups(self):
        return [[p] for m in self.norm_list for p in (m.weight, m.bias)]

    def nacf_bind(self, flat, rt):
        self._packs = [flat.pack([m.weight], [m.bias]) for m in self.norm_list]

    def forward(self, encoder_outputs, encoder_hiddens=None):
        if not isinstance(encoder_outputs, (list, tuple)):
            encoder_outputs = [encoder_outputs]
        if not self.norm_list:
            return torch.cat(list(encoder_outputs), dim=1), encoder_hiddens  # pure data movement
        assert len(encoder_outputs) == len(self.norm_list)
        params = [p for m in self.norm_list for p in (m.weight, m.bias)]
        if not self.is_bn:
            cfg = dict(packs=self._packs, eps=self.norm_list[0].eps)
            return LNConcatFn.apply(cfg, len(encoder_outputs), *encoder_outputs, *params), encoder_hiddens
        mods = [dict(pack=pk, running_mean=m.running_mean, running_var=m.running_var, nbt=m.num_batches_tracked)
                for pk, m in zip(self._packs, self.norm_list)]
        cfg = dict(mods=mods, training=self.training, momentum=self.norm_list[0].momentum, eps=self.norm_list[0].eps,
                   sync=self._sync if self.sync_bn else None)
        params = [p for m in self.norm_list for p in (m.weight, m.bias)]
        out = BNConcatFn.apply(cfg, len(encoder_outputs), *encoder_outputs, *params)
        return out, encoder_hiddens
