"""Model factory / plugin registry -- drop-in for the reference's
`models.get_model(opt)` (models/__init__.py:25-94): encoder / decoder classes
are resolved BY NAME from opt['encoder'] / opt['decoder'] against each module's
`__all__`, predictors from opt['crit'], and sub-modules are constructed in the
reference's order so the default initialisers draw the same random numbers."""
import torch.nn as nn

from ..config import Constants
from ..opts import complete_opt
from . import Decoder, Encoder, Predictor
from .joint_representation import Joint_Representaion_Learner
from .seq2seq import Seq2Seq


def _resolve(module, name, key_name):
    if name not in module.__all__:
        raise ValueError('We can not find {} in models/{}.py (supported: {})'.format(name, key_name, module.__all__))
    return getattr(module, name)


def get_encoder(opt, input_size):
    return _resolve(Encoder, opt['encoder'], 'Encoder')(opt)


def get_joint_representation_learner(opt):
    if opt.get('no_joint_representation_learner', False):
        return None
    return Joint_Representaion_Learner([opt['dim_hidden']] * len(opt['modality']), opt)


def get_auxiliary_task_predictor(opt):
    supported = [item[10:] for item in dir(Predictor) if 'Predictor_' in item]
    layers = []
    for crit_name in opt['crit']:
        if crit_name in supported:
            layers.append(getattr(Predictor, 'Predictor_%s' % crit_name)(opt, key_name=Constants.mapping[crit_name][0]))
    return None if not layers else Predictor.Auxiliary_Task_Predictor(layers)


def get_decoder(opt):
    return _resolve(Decoder, opt['decoder'], 'Decoder')(opt)


def get_model(opt):
    assert 'vocab_size' in opt, "opt['vocab_size'] must be set (train.py:73 does it from the corpus)"
    assert not opt.get('use_preEncoder', False)
    opt = complete_opt(opt)
    if opt.get('gemm_mode') is not None:
        # arithmetic of every GEMM (process-wide, see nacf_hip.h NACF_GEMM_*): 'f32' | 'bf16x3' (exact) | 'bf16'
        from ..runtime import ops
        ops.set_gemm_mode(opt['gemm_mode'])
    sizes = {'i': opt['dim_i'], 'm': opt['dim_m'], 'a': opt['dim_a'], 'o': opt['dim_o']}
    input_size = [sizes[c] for c in opt['modality'].lower()]
    encoder = get_encoder(opt, input_size)
    joint_representation_learner = get_joint_representation_learner(opt)
    auxiliary_task_predictor = get_auxiliary_task_predictor(opt)
    decoder = get_decoder(opt)
    tgt_word_prj = nn.Linear(opt['dim_hidden'], opt['vocab_size'], bias=False)
    return Seq2Seq(opt=opt, preEncoder=None, encoder=encoder,
                   joint_representation_learner=joint_representation_learner,
                   auxiliary_task_predictor=auxiliary_task_predictor, decoder=decoder, tgt_word_prj=tgt_word_prj)
