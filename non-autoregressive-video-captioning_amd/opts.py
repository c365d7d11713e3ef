"""Option dictionaries for the hot path: the defaults of the reference's flag
system that shape the model / decode (opts.py:23-50,53-62,87-103,116-129) and
its method / dataset overlays (opts.py:148-213), without argparse or file-system
checks (no teacher-checkpoint assertion, no ./config relative open)."""
import os

import yaml

from .config import Constants

_HERE = os.path.dirname(os.path.abspath(__file__))

DEFAULTS = dict(
    dataset='MSRVTT', modality='mi', method='',
    encoder='Encoder_HighWay', decoder='BertDecoder', decoding_type='ARFormer', fusion='temporal_concat',
    dim_hidden=512, num_hidden_layers_decoder=1, num_attention_heads=8, intermediate_size=2048,
    hidden_act='gelu_new', hidden_dropout_prob=0.5, attention_probs_dropout_prob=0.0, max_len=30,
    layer_norm_eps=1e-5, watch=0, pos_attention=False, enhance_input=2, with_layernorm=False,
    with_category=False, num_category=20, encoder_dropout=0.5, no_encoder_bn=False, norm_type='bn',
    tie_weights=False, seed=0, learning_rate=5e-4, decay=0.9, minimum_learning_rate=5e-5, n_warmup_steps=0,
    optim='adam', grad_clip=5.0, weight_decay=5e-4, epochs=50, batch_size=64, beta=[0, 1],
    visual_word_generation=False, demand=['VERB', 'NOUN'], nv_weights=[0.8, 1.0],
    beam_size=1, beam_alpha=1.0, topk=1, paradigm='mp', length_beam_size=6, iterations=5, q=1, q_iterations=1,
    use_ct=False, length_bias=0, crit=['lang'], crit_name=['Cap Loss'], crit_scale=[1.0],
    n_frames=8, random_type='segment_random', load_feats_type=1, dim_a=1, dim_m=2048, dim_i=2048, dim_o=1,
    # evaluation / checkpointing defaults of the training driver (opts.py:75-85; read by misc/run.py and misc/logger.py)
    start_eval_epoch=0, save_checkpoint_every=1, tolerence=1000, k_best_model=1, standard=['METEOR', 'CIDEr'],
)

# keys that configure THIS runtime, not the model: never written into a checkpoint's `settings` (the reference's
# loaders would carry them along, and a reloaded model must return the reference's forward contract by default)
RUNTIME_KEYS = ('fused_loss', 'hipgraph', 'decode_graph', 'gemm_mode', 'sync_bn', 'encoder_joint_streams', 'ar_last_slot_only')


def persistable(opt):
    """a copy of `opt` without the runtime-only keys (what goes into checkpoint['settings'])"""
    return {k: v for k, v in opt.items() if k not in RUNTIME_KEYS}


def load_methods():
    with open(os.path.join(_HERE, 'config', 'methods.yaml')) as f:
        return yaml.safe_load(f)


def make_opt(method='', dataset='MSRVTT', default=False, **overrides):
    """Build an `opt` dict the way `opts.parse_opt()` + `vars()` does upstream."""
    opt = dict(DEFAULTS)
    opt.update(dataset='Youtube2Text' if dataset.lower() == 'msvd' else dataset, method=method)
    assert opt['dataset'] in ('Youtube2Text', 'MSRVTT')
    pre = {k: overrides[k] for k in overrides}
    if default:  # opts.py:161-169
        if opt['dataset'] == 'Youtube2Text':
            opt.update(beta=[0, 1], max_len=20, with_category=False)
        else:
            opt.update(beta=[0.35, 0.9], max_len=30, with_category=True)
    opt.update(pre)
    if opt['dataset'] == 'Youtube2Text':
        assert not opt['with_category'], 'no category information in Youtube2Text (MSVD)'
    if method:  # opts.py:176-183
        methods = load_methods()
        assert method in methods, 'unknown method %s' % method
        opt.update(methods[method])
    if opt['decoding_type'] == 'NARFormer':  # opts.py:185-189
        opt.update(crit=['lang', 'length'], crit_name=['Cap Loss', 'Length Loss'], crit_scale=[1.0, 1.0])
    opt['crit_key'] = [Constants.mapping[c.lower()] for c in opt['crit']]
    if default:  # opts.py:191-213, minus the teacher-checkpoint assertion
        if opt['decoding_type'] == 'NARFormer':
            if opt['visual_word_generation']:
                opt.update(use_ct=True, nv_weights=[0.8, 1.0])
            opt.update(enhance_input=2, length_beam_size=6, iterations=5,
                       beam_alpha=1.35 if opt['dataset'] == 'MSRVTT' else 1.0)
        else:
            opt.update(beam_size=5, beam_alpha=1.0)
    opt.update(pre)
    return opt


def complete_opt(opt):
    """fill keys a hand-written opt dict may lack (never overrides)"""
    out = dict(DEFAULTS)
    out.update(opt)
    if 'crit_key' not in out:
        out['crit_key'] = [Constants.mapping[c.lower()] for c in out['crit']]
    return out
