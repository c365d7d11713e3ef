"""CPU restatement of the reference's batch construction: frame sampling, caption padding,
masked-LM source/target pairs, visual-word targets, length-target distributions.  TEST INFRASTRUCTURE ONLY (same
rules as nacf_oracle.py): only tests/ and the generators under oracle/ import it.  Pinned against the reference's
own dataloader methods by oracle/make_golden.py::data_case (fixture tests/golden/tiny_data.npz).

All line numbers refer to dataloader.py of the reference checkout."""
import numpy as np

PAD, UNK, BOS, EOS, MASK, VIS = 0, 1, 2, 3, 4, 5          # config/Constants.py:1-6
BE_VERBS = ('is', 'are', 'was', 'were', 'be')             # dataloader.py:402


def resampling(source_length, target_length):
    """:20-21 -- stretch a short clip: round(i * (S-1) / (T-1))"""
    return [round(i * (source_length - 1) / (target_length - 1)) for i in range(target_length)]


def frame_bounds(n_total_frames, n_frames):
    """:27 -- int(np.linspace(0, total, n+1))"""
    return [int(v) for v in np.linspace(0, n_total_frames, n_frames + 1)]


def get_frame_ids(n_total_frames, n_frames, random_type, rng=None):
    """:24-37.  'equally_sampling' = the middle of each of n segments; 'segment_random' = one uniform draw per segment
    (the reference uses numpy's global generator; pass the RandomState to reproduce a stream)."""
    bound = frame_bounds(n_total_frames, n_frames)
    idx = []
    for i in range(n_frames):
        if random_type == 'equally_sampling':
            idx.append((bound[i] + bound[i + 1]) // 2)
        elif random_type == 'segment_random':
            idx.append(int((rng or np.random).randint(bound[i], bound[i + 1])))
        else:
            raise NotImplementedError(random_type)
    return sorted(idx)


def select_frames(n_source, n_frames, max_seq_len, load_feats_type, random_type, rng=None):
    """:296-315 -- which rows of a [n_source, D] feature matrix a sample keeps"""
    if load_feats_type == 1:
        if n_source >= n_frames:
            return get_frame_ids(n_source, n_frames, random_type, rng)
        return resampling(n_source, max_seq_len)
    if load_feats_type == 2:
        return resampling(n_source, max_seq_len) if n_source < max_seq_len else list(range(n_source))
    raise NotImplementedError('load_feats_type 0 draws frame ids once per sample for all modalities (:231-236)')


def padding(seq, max_len, add_eos=True):
    """:317-327"""
    res = list(seq)
    if len(res) > max_len:
        res = res[:max_len]
        if add_eos:
            res[-1] = EOS
    else:
        res += [PAD] * (max_len - len(res))
    return res


def mlm_num_masks_range(n, beta):
    """:349-353 -> [low, high) of the number of masked slots of an n-word sentence (None: nothing is masked)"""
    if 1 >= n:
        return None
    low = max(int(n * beta[0]), 1)
    high = max(int(n * beta[1]), 1)
    if high == low:
        high += 1
    return low, high


def source_target_mlm(sent, beta, max_len, train, rng=None, ind=None):
    """:346-380.  sent = caption without <bos>/<eos>.  Training: a random subset `ind` of the slots (size uniform in
    [low, high), positions without replacement) becomes <mask> in the source and keeps its word in the target, every
    other target slot is <pad>; evaluation: everything is masked and the target is the sentence."""
    assert sent[0] != BOS and sent[-1] != EOS
    n = len(sent)
    src, tgt = list(sent), [PAD] * n
    if train:
        if ind is None:
            rg = mlm_num_masks_range(n, beta)
            ind = [] if rg is None else rng.choice(n, size=rng.randint(rg[0], rg[1]), replace=False)
        for i in ind:
            src[i] = MASK
            tgt[i] = sent[i]
    else:
        src = [MASK if t != PAD else t for t in sent]
        tgt = list(sent)
    return padding(src, max_len, add_eos=False), padding(tgt, max_len, add_eos=False)


def source_target_visual_word(target, pos_tag, demanded, is_be, max_len, narformer, train):
    """:382-425.  target / pos_tag include <bos>/<eos>; demanded[tag] and is_be[word] are boolean look-ups
    (itop[tag] in opt['demand'], itow[word] in BE_VERBS)."""
    if not train:
        return [0], [0]
    n = len(target) - 2
    src1 = padding([VIS] * (n if narformer else len(target)), max_len, add_eos=not narformer)
    tgt1 = [MASK] * n
    for i in range(n):
        if demanded[pos_tag[i + 1]] and not is_be[target[i + 1]]:
            tgt1[i] = target[i + 1]
    if narformer:
        tgt1 = padding(tgt1, max_len, add_eos=False)
    else:
        tgt1 = padding([target[0]] + tgt1 + [EOS], max_len, add_eos=True)
    return src1, tgt1


def make_source_target(target, pos_tag, opt, demanded, is_be, train, rng=None, ind=None):
    """:329-344 -> dict(tokens, labels[, tokens_1, labels_1])"""
    narformer = opt['decoding_type'] == 'NARFormer'
    if narformer:
        src, tgt = source_target_mlm(target[1:-1], opt.get('beta', [0, 1]), opt['max_len'], train, rng, ind)
    else:
        src = padding(target, opt['max_len'], add_eos=True)
        tgt = list(src)
    out = {'tokens': src, 'labels': tgt}
    if opt.get('visual_word_generation', False):
        out['tokens_1'], out['labels_1'] = source_target_visual_word(target, pos_tag, demanded, is_be, opt['max_len'],
                                                                     narformer, train)
    return out


def length_target(length_info_vid, max_len):
    """:166-175 -- normalised histogram of the caption lengths of one video, cut / zero-padded to max_len"""
    lt = list(length_info_vid)[:max_len]
    lt += [0] * (max_len - len(lt))
    return np.array(lt) / sum(lt)
