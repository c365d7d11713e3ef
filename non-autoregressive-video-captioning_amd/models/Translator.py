"""Decode plugin surface (reference: models/Translator.py:12-186): same
constructor and `translate_batch` signature / return shapes.
NARFormer -> `decoding.generate` (device-side coarse-to-fine mask-predict);
ARFormer  -> batched beam search (`models.Beam`) for the ARB / ARB2 comparator."""
import torch

from .. import decoding


class Translator(object):
    def __init__(self, model, opt, device=torch.device('cuda'), teacher_model=None, dict_mapping={}):
        self.model = model
        self.model.eval()
        self.opt = opt
        self.device = device
        self.teacher_model = teacher_model
        if teacher_model is not None:
            teacher_model.eval()
        self.dict_mapping = dict_mapping
        self.length_bias = opt.get('length_bias', 0)

    def translate_batch_ARFormer(self, encoder_outputs, category):
        from .Beam import beam_search
        with torch.no_grad():
            return beam_search(self.model, self.opt, encoder_outputs, category)

    def translate_batch_NARFormer(self, encoder_outputs, teacher_encoder_outputs, category, tgt_tokens, tgt_vocab,
                                  **kwargs):
        with torch.no_grad():
            return decoding.generate(opt=self.opt, model=self.model, teacher_model=self.teacher_model,
                                     encoder_outputs=encoder_outputs,
                                     teacher_encoder_outputs=teacher_encoder_outputs, category=category,
                                     tgt_tokens=tgt_tokens, tgt_vocab=tgt_vocab, dict_mapping=self.dict_mapping,
                                     length_bias=self.length_bias, **kwargs)

    def translate_batch(self, encoder_outputs, category, tgt_tokens, tgt_vocab, teacher_encoder_outputs=None,
                        **kwargs):
        if self.opt['decoding_type'] == 'NARFormer':
            return self.translate_batch_NARFormer(encoder_outputs, teacher_encoder_outputs, category, tgt_tokens,
                                                  tgt_vocab, **kwargs)
        if self.opt['decoding_type'] == 'ARFormer':
            return self.translate_batch_ARFormer(encoder_outputs, category)
        raise NotImplementedError('nacf_amd: decoding_type %s is not built' % self.opt['decoding_type'])
