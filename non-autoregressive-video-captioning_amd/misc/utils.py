"""Checkpoint contract of the reference.

A reference checkpoint is `torch.save({'epoch', 'state_dict', 'validate_result', 'settings'}, path)`
(misc/run.py:334-339); `settings` is the option dict `get_model` was built from.  Because this package keeps the
reference's parameter names, shapes and option keys, the authors' released `best.pth.tar` files load unchanged:

  * load_model_and_opt(path, device[, return_other_info])     misc/utils.py:54-63
  * load_satisfied_weights(model, path, str_mapping, ...)     misc/utils.py:158-192  (teacher / pre-training init:
        run.py:275-283 maps 'decoder.bert.' -> 'decoder.' to start a NACF student from an ARB checkpoint)
  * save_checkpoint(state, is_best, filepath, ...)            misc/utils.py:195-202

Tensors are read on the host and copied into the model's flat parameter buffer by `load_state_dict`.
"""
import os
import shutil

import torch


def _read(checkpoint_path):
    # reference checkpoints carry plain containers only (tensors, numbers, strings, lists, dicts)
    return torch.load(checkpoint_path, map_location='cpu', weights_only=False)


def load_model_and_opt(checkpoint_path, device, return_other_info=False):
    from ..models import get_model
    checkpoint = _read(checkpoint_path)
    opt = checkpoint['settings']
    model = get_model(opt)
    model.load_state_dict(checkpoint['state_dict'])
    model.to(device)
    if not return_other_info:
        return model, opt
    other = {k: v for k, v in checkpoint.items() if k != 'state_dict'}
    return model, opt, other


def load_satisfied_weights(model, checkpoint_path, str_mapping=None, skip_keys=(), strict=False):
    """Copy every entry of the checkpoint that has a counterpart in `model`.  A model key containing one of the
    `str_mapping` keys is looked up under the name with that substring replaced (first matching rule wins);
    keys in `skip_keys` keep the model's value; a missing counterpart raises only when `strict`."""
    str_mapping = dict(str_mapping or {})
    source = _read(checkpoint_path)['state_dict']
    current = model.state_dict()
    merged, hits = {}, 0
    for name, value in current.items():
        merged[name] = value
        if name in skip_keys:
            continue
        lookup = name
        for old, new in str_mapping.items():
            if old in name:
                lookup = name.replace(old, new)
                break
        if lookup in source:
            merged[name] = source[lookup]
            hits += 1
        elif strict:
            raise AssertionError('key {}/{} can not be found in the checkpoint'.format(name, lookup))
    print('Successfully loading {}/{} parameters'.format(hits, len(merged)))
    model.load_state_dict(merged)
    return model


def save_checkpoint(state, is_best, filepath='./', filename='checkpoint.pth.tar', best_model_name='best.pth.tar'):
    os.makedirs(filepath, exist_ok=True)
    target = os.path.join(filepath, filename)
    torch.save(state, target)
    if is_best:
        shutil.copyfile(target, os.path.join(filepath, best_model_name))
