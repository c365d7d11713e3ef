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
    from ..opts import persistable
    opt = persistable(checkpoint['settings'])     # runtime-only keys never configure a reloaded model
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


# ---- text helpers used by run_eval (reference: misc/utils.py:11-147) ----------------------------------------------
def set_seed(seed=2019):
    import random
    import numpy as np
    random.seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def to_sentence(hyp, vocab, break_words=(3, 0), skip_words=()):
    """ids -> words joined by blanks, up to the first <eos>(3) / <pad>(0)  (misc/utils.py:21-30)"""
    words = []
    for wid in hyp:
        if wid in skip_words:
            continue
        if wid in break_words:
            break
        words.append(vocab[wid])
    return ' '.join(words)


def remove_repeat_n_grame(sent, n):
    """Drop ONE immediate repetition of an n-gram (second copy adjacent to, or one word after, the FIRST occurrence
    of that n-gram; the in-between word goes too) and return (words, False); (sent, True) when nothing repeats
    (misc/utils.py:66-81)."""
    first_at = {}
    for i in range(len(sent) - n + 1):
        gram = ' '.join(sent[i:i + n])
        if gram not in first_at:
            first_at[gram] = i
            continue
        gap = i - first_at[gram] - n
        if gap == 0 or gap == 1:
            return sent[:i - gap] + sent[i + n:], False
    return sent, True


def duplicate(sent):
    """de-duplication post-process of NA captions, 4-grams first (misc/utils.py:84-98): returns (sentence, report)"""
    words = sent.split(' ')
    removed = {}
    for n in (4, 3, 2, 1):
        while True:
            words, clean = remove_repeat_n_grame(words, n)
            if clean:
                break
            removed[n] = removed.get(n, 0) + 1
    return ' '.join(words), '\t'.join('%d-gram: %d' % (n, removed.get(n, 0)) for n in (1, 2, 3, 4))


def _count_ngrams(sentences, n):
    grams = {}
    for words in sentences:
        for j in range(len(words) - n + 1):
            g = ' '.join(words[j:j + n])
            grams[g] = grams.get(g, 0) + 1
    return grams


def analyze_length_novel_unique(gt_data, data, vocab, splits, n=1, calculate_novel=True):
    """(average length, novel ratio, unique ratio, vocabulary usage, n-gram counts, #distinct 4-grams) of predicted
    captions `data` = {vid: [{'caption': str}, ...]}; novel = not among the training captions (misc/utils.py:101-146)"""
    hyps = [item['caption'] for k in data for item in data[k]]
    split_hyps = [h.split(' ') for h in hyps]
    count = len(hyps)
    distinct = set(hyps)
    novel = 0
    if calculate_novel:
        train = set()
        for i in splits['train']:
            for cap in gt_data['video%d' % int(i)]:
                train.add(' '.join(vocab[w] for w in cap[1:-1]))
        novel = sum(1 for s in distinct if s not in train)
    grams = _count_ngrams(split_hyps, n)
    return (sum(len(w) for w in split_hyps) / count, novel / count, len(distinct) / count, len(grams), grams,
            len(_count_ngrams(split_hyps, 4)))
