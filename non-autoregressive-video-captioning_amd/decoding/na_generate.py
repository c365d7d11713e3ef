"""NA decoding driver (reference: decoding/na_generate.py:14-135).

Same entry point and option keys as the reference `decoding.generate`; what
changed is where the work runs: the length beam, canvas, candidate scoring and
best-candidate gather are device kernels, the visual memory is addressed by
`row // lbs` instead of being repeated lbs times (misc/utils.py:205-229), and
its cross-attention K|V projection is computed once per video.  The single host
read per batch is the canvas width (`beam.max().item()`, na_generate.py:37),
kept so the returned hypotheses have exactly the reference's shape; it moves
behind the work, which replays from one hipGraph (`_generate_graphed`,
opt['decode_graph'] = 'auto' | 'on' | 'off'): mask-predict as is, left-to-right
and easy-first with their pass-count upper bounds (Algorithm_Base.static_passes).
"""
import torch

from ..runtime import ops
from ..runtime.functional import MeanTimeFn
from .algorithms import DecodeContext, EasyFirst, Left2Right, MaskPredict

algorithms_mapping = {'mp': MaskPredict, 'l2r': Left2Right, 'ef': EasyFirst}


def generate(opt, model, teacher_model, encoder_outputs, teacher_encoder_outputs, category, tgt_tokens, tgt_vocab,
             dict_mapping, length_bias, **kwargs):
    paradigm = opt.get('paradigm', 'mp')
    assert paradigm in ('mp', 'l2r', 'ef')
    # opt['load_generated_captions'] (na_generate.py:25-26): the length beam is centred on the given captions' lengths
    gold = tgt_tokens if opt.get('load_generated_captions', False) else None
    if gold is not None and not torch.is_tensor(gold):
        raise ValueError('nacf_amd: load_generated_captions needs tgt_tokens (the captions whose lengths seed the beam)')
    if kwargs.get('output_attentions', False) or opt.get('example', ''):
        raise NotImplementedError('nacf_amd: attention collection / example mode of generate() is not built')
    mode = opt.get('decode_graph', 'auto')
    assert mode in ('auto', 'on', 'off')
    # mask-predict has a fixed launch sequence once the canvas width is fixed: replay it from a hipGraph.  'l2r' / 'ef' read a
    # slot count on the host between passes (algorithms.py:275-418); under capture they run their pass-count upper bounds
    # instead (Algorithm_Base.static_passes) -- unless the per-pass results are collected, whose number the extra passes
    # would change.  (A vocabulary remap table is uploaded per call: not graphed.)
    per_pass = opt.get('collect_best_candidate_iterative_results', False) and not opt.get('collect_last', False)
    if mode != 'off' and not dict_mapping and (paradigm == 'mp' or not per_pass):
        out = _generate_graphed(opt, model, teacher_model, encoder_outputs, teacher_encoder_outputs, category,
                                tgt_vocab, length_bias, mode, gold)
        if out is not None:
            return out
    hyp, lprobs, _, _ = _generate(opt, model, teacher_model, encoder_outputs, teacher_encoder_outputs, category, tgt_vocab,
                                  dict_mapping, length_bias, None, gold)
    return hyp, lprobs


def _enc_tensors(enc):
    e = enc['enc_output']
    return e[0] if isinstance(e, list) else e, enc.get('pred_length'), enc.get('_pooled_memory')


def _generate_graphed(opt, model, teacher_model, enc, t_enc, category, tgt_vocab, length_bias, mode, gold=None):
    """The whole decode of one batch -- length beam, canvas, T(+1) decoder passes with fused projection/argmax,
    re-masking, optional teacher scoring, candidate selection -- as ONE hipGraph replay over static inputs.  Inside
    the graph the canvas is max_len-1 slots wide (the length beam's upper clamp, na_generate.py:133): the extra
    slots are <pad> in every candidate, so they are skipped by the live-row GEMMs and change no value; the result
    is cut back to the reference's width (`beam.max()`) with the single host read the reference performs too.
    mode 'auto': a batch geometry is captured the second time it is seen (a one-off decode or the ragged last batch
    of an evaluation never pays for a capture); 'on': at once."""
    e, pl, pooled = _enc_tensors(enc)
    if not e.is_cuda:
        return None
    te = _enc_tensors(t_enc) if (teacher_model is not None and t_enc is not None) else None
    keys = ('paradigm', 'use_ct', 'iterations', 'length_beam_size', 'beam_alpha', 'masking_decision',
            'no_candidate_decision', 'collect_best_candidate_iterative_results', 'collect_last',
            'not_only_best_candidate', 'q', 'q_iterations')
    flat = getattr(model, 'flat', None)          # the graph holds raw pointers into the parameter buffer
    key = (None if flat is None else (flat.data.data_ptr(), flat.image_epoch), ops.gemm_mode(), tuple(e.shape), tuple(pl.shape), None if category is None else tuple(category.shape), int(length_bias),
           id(teacher_model) if te is not None else None, None if te is None else tuple(te[0].shape),
           tuple(str(opt.get(k)) for k in keys), None if gold is None else tuple(gold.shape))
    cache = model.__dict__.setdefault('_nacf_decode_graphs', {})
    entry = cache.get(key)
    if entry is None:
        if mode == 'auto' and not cache.setdefault(('seen', key), False):
            cache[('seen', key)] = True
            return None
        W = pl.shape[1] - 1
        st = dict(e=e.clone(), pl=pl.clone(), pooled=None if pooled is None else pooled.clone(),
                  cat=None if category is None else category.clone(),
                  gold=None if gold is None else gold.to(device=e.device, dtype=torch.int64).contiguous().clone(),
                  te=None if te is None else [None if x is None else x.clone() for x in te])

        def run():
            s_enc = {'enc_output': st['e'], 'pred_length': st['pl']}
            if st['pooled'] is not None:
                s_enc['_pooled_memory'] = st['pooled']
            s_t = None
            if st['te'] is not None:
                s_t = {'enc_output': st['te'][0]}
                if st['te'][2] is not None:
                    s_t['_pooled_memory'] = st['te'][2]
            return _generate(opt, model, teacher_model, s_enc, s_t, st['cat'], tgt_vocab, {}, length_bias, W, st['gold'])
        run()                                   # launch by launch once: workspaces, mask-count tables
        dev = e.device
        torch.cuda.synchronize(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side, capture_error_mode='thread_local'):
                outs = run()
        torch.cuda.current_stream(dev).wait_stream(side)
        entry = cache[key] = (graph, st, outs)
    graph, st, (hyp, lprobs, beam_max, status) = entry
    st['e'].copy_(e)
    st['pl'].copy_(pl)
    if pooled is not None:
        st['pooled'].copy_(pooled)
    if category is not None:
        st['cat'].copy_(category)
    if gold is not None:
        st['gold'].copy_(gold)
    if te is not None:
        for dst, src in zip(st['te'], te):
            if src is not None:
                dst.copy_(src)
    graph.replay()
    Lp = int(beam_max.item())                   # the reference's one host read (na_generate.py:37), after the work
    if status is not None:                      # EasyFirst under a fixed pass count: open slots before / after the last pass
        before, after = (int(v) for v in status.tolist())
        if after != 0 and after != before:      # still filling (a prediction was the <mask> id): the host-driven loop decides
            return None
    out_l = None if lprobs is None else tuple(x[..., :Lp].clone() for x in lprobs)
    return hyp[:, :Lp].clone(), out_l


def _generate(opt, model, teacher_model, encoder_outputs, teacher_encoder_outputs, category, tgt_vocab, dict_mapping,
              length_bias, fixed_width, gold=None):
    paradigm = opt.get('paradigm', 'mp')
    algorithm = algorithms_mapping[paradigm](opt, dict_mapping, tgt_vocab)
    lbs = opt['length_beam_size']
    beam_alpha = opt.get('beam_alpha', 1.0)
    pred_length = encoder_outputs['pred_length'].contiguous()
    dev = pred_length.device
    B, max_len = pred_length.shape
    lbs = min(lbs, max_len)

    beam = torch.empty(B, lbs, dtype=torch.int32, device=dev)
    beam_max = torch.empty(1, dtype=torch.int32, device=dev)
    if gold is not None:        # no length_bias here, as upstream (na_generate.py:118-122)
        gold = gold.to(device=dev, dtype=torch.int64).contiguous()
        ops.length_beam_gold(gold, max_len, lbs, beam, beam_max)
    else:
        ops.length_beam(pred_length, lbs, int(length_bias), beam, beam_max)
    Lp = int(beam_max.item()) if fixed_width is None else int(fixed_width)
    R = B * lbs
    tokens = torch.empty(R, Lp, dtype=torch.int64, device=dev)
    if gold is not None:        # the canvas starts from the given captions (na_generate.py:42-45)
        if fixed_width is None and gold.shape[1] < Lp:
            raise ValueError('nacf_amd: tgt_tokens is %d wide, the longest length candidate is %d' % (gold.shape[1], Lp))
        ops.canvas_init_gold(beam, gold, R, lbs, Lp, tokens)
    else:
        ops.canvas_init(beam, R, Lp, tokens)

    enc_output = encoder_outputs['enc_output']
    if isinstance(enc_output, list):
        enc_output = enc_output[0]
    pooled = encoder_outputs.get('_pooled_memory')
    if pooled is None:
        pooled = MeanTimeFn.apply(enc_output)
    ctx = DecodeContext(model, enc_output, category, pooled, lbs)
    teacher_ctx = None
    if teacher_model is not None and teacher_encoder_outputs is not None:
        t_enc = teacher_encoder_outputs['enc_output']
        t_pool = teacher_encoder_outputs.get('_pooled_memory')
        if t_pool is None:
            t_pool = MeanTimeFn.apply(t_enc)
        teacher_ctx = DecodeContext(teacher_model, t_enc, category, t_pool, lbs)

    tokens, probs, teacher_probs, collect_results = algorithm.generate(ctx, teacher_ctx, tokens)

    hypotheses = torch.empty(B, Lp, dtype=torch.int64, device=dev)
    best = torch.empty(B, dtype=torch.int32, device=dev)
    ops.best_candidate(tokens, probs, teacher_probs, beam, beam_alpha, B, lbs, Lp, hypotheses, best, None)

    lprobs = None  # "For speedup" upstream as well (na_generate.py:78)
    if collect_results[0]:
        sents, scores, _ = collect_results
        if not opt.get('not_only_best_candidate', False) and not opt.get('collect_last', False):
            idx = best.long().view(B, 1, 1).expand(B, 1, Lp)
            sents = [s.view(B, lbs, Lp).gather(1, idx).squeeze(1) for s in sents]
            scores = [s.view(B, lbs, Lp).gather(1, idx).squeeze(1) for s in scores]
        lprobs = (torch.stack(sents, dim=1), torch.stack(scores, dim=1))
    return hypotheses, lprobs, beam_max, algorithm.status
