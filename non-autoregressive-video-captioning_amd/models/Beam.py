"""Autoregressive beam search for ARB / ARB2 (the config-5 comparator).

Reference: one Python `Beam` object per instance (models/Beam.py:5-169) driven
by `Translator.translate_batch_ARFormer` (models/Translator.py:94-161), with
host-side hypothesis reconstruction, per-element `.item()` loops and
`index_select` batch shrinking every step.  Here the whole batch advances in
one `nacf_beam_step` launch per step: flat top-k over beam x vocab with the
<eos> masking rule, back-pointer re-ordering, finished-list bookkeeping and the
done flags all stay on the device; the host reads one int32 (instances still
active) per step.  Like the reference, every step re-runs the decoder on the
whole prefix (no KV cache) and takes the last position -- but only THAT
position's hidden state is ever read (Translator.py:111: `dec_output[:, -1, :]`),
so the last decoder layer projects keys / values for the whole prefix and runs
everything else -- query projection, one-query attention cores, both output
projections, the FFN -- densely on the last slot of every hypothesis
(BertLayer._run_last; opt['ar_last_slot_only'] = False restores the full pass).

Returns what the reference returns: (list[B] of list[n_best] of token lists,
list[B] of list[n_best] of length-normalised scores).
"""
import torch

from ..config import Constants
from ..runtime import ops


def beam_search(model, opt, encoder_outputs, category):
    n_bm = int(opt['beam_size'])
    max_len = int(opt['max_len'])
    topk = int(opt.get('topk', 1))
    alpha = opt.get('beam_alpha', 1.0)
    want = max(n_bm, topk)
    enc_output = encoder_outputs['enc_output']
    if isinstance(enc_output, list):
        enc_output = enc_output[0]
    dev = enc_output.device
    B = enc_output.shape[0]
    R = B * n_bm
    W, bias = model.tgt_word_prj.weight, model.tgt_word_prj.bias
    V = W.shape[0]
    memory_kv = model.decoder.project_memory(enc_output)

    seqs = torch.zeros(B, n_bm, max_len, dtype=torch.int64, device=dev)     # PAD
    seqs[:, 0, 0] = Constants.BOS                                          # only beam 0 starts with <bos> (Beam.py:27-29)
    scores = torch.zeros(B, n_bm, dtype=torch.float32, device=dev)
    fin_scores = torch.zeros(B, want, dtype=torch.float32, device=dev)
    fin_len = torch.zeros(B, want, dtype=torch.int32, device=dev)
    fin_tokens = torch.zeros(B, want, max_len, dtype=torch.int64, device=dev)
    fin_count = torch.zeros(B, dtype=torch.int32, device=dev)
    done = torch.zeros(B, dtype=torch.int32, device=dev)
    n_active = torch.zeros(1, dtype=torch.int32, device=dev)
    buf = torch.empty(R, ops.vocab_ld(V), dtype=torch.float32, device=dev)
    logits = buf[:, :V]

    subset = opt.get('ar_last_slot_only', True)
    host = model.__dict__.get('_nacf_beam_host')          # two pinned counters + events, kept with the model
    if host is None:
        host = ([torch.ones(1, dtype=torch.int32).pin_memory() for _ in range(2)], [torch.cuda.Event() for _ in range(2)])
        model.__dict__['_nacf_beam_host'] = host
    host_cnt, ev = host
    for c_ in host_cnt:
        c_.fill_(1)
    for t in range(1, max_len):
        tokens = seqs[:, :, :t].reshape(R, t).contiguous()
        out = model.decoder(tokens, enc_output=enc_output, category=category, decoding_type='ARFormer',
                            row_map=('div', n_bm), memory_kv=memory_kv, last_slot_only=subset)
        h = out[0]
        if isinstance(h, list):
            h = h[-1]
        last = h[:, -1, :]                                                  # pick the last step (Translator.py:111)
        ops.linear_fwd(last, W, logits, ops.Epi(bias=bias))
        ops.vocab_logsoftmax_fwd(logits, V, None, None, None, None)
        ops.beam_step(logits, V, t, max_len, want, seqs, scores, fin_scores, fin_len, fin_tokens, fin_count, done,
                      n_active)
        # all instances reached <eos> (Translator.py:153-154).  The count travels to the host asynchronously and is looked at
        # ONE step late: the device never waits for the host's read (a blocking .item() per step idled it for a host round
        # trip, 19 times per batch), and the one step that may run after everything has finished changes nothing -- finished
        # instances are skipped by nacf_beam_step and their results are already final.
        host_cnt[t % 2].copy_(n_active, non_blocking=True)
        ev[t % 2].record()
        if t > 1:
            ev[(t - 1) % 2].synchronize()
            if int(host_cnt[(t - 1) % 2]) == 0:
                break

    # sort_finished (Beam.py:123-130): score / len^alpha, stable descending sort, n_best hypotheses
    f_sc, f_len, f_tok, f_cnt = fin_scores.tolist(), fin_len.tolist(), fin_tokens.tolist(), fin_count.tolist()
    batch_hyp, batch_scores = [], []
    n_best = topk   # upstream quirk kept: n_best shrinks monotonically across instances (Translator.py:84-92)
    for b in range(B):
        items = [[f_sc[b][i] / (f_len[b][i] ** alpha), f_len[b][i], i] for i in range(f_cnt[b])]
        items.sort(key=lambda a: -a[0])
        n_best = min(n_best, len(items))
        batch_scores.append([it[0] for it in items[:n_best]])
        batch_hyp.append([f_tok[b][it[2]][1:1 + it[1]] for it in items[:n_best]])
    return batch_hyp, batch_scores
