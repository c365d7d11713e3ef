"""torch.autograd.Function wrappers: each forward/backward is a short sequence
of libnacf_hip launches.  Autograd is used only to chain them (plumbing).

Parameter gradients are written by the kernels straight into the model's flat
gradient buffer (views handed over in ``Pack`` objects, beta = 1 accumulate),
so the optimiser and the RCCL all-reduce see one contiguous fp32 bucket; the
Functions therefore return ``None`` for parameter inputs, which are passed
only to connect the autograd graph.
"""
from __future__ import annotations

from typing import List, Optional

import os

import torch
from torch.autograd import Function

from . import lib as L
from . import ops

Tensor = torch.Tensor


class Pack:
    """A (possibly packed) weight/bias pair living in the flat buffers:
    ``w``/``b`` are data views, ``gw``/``gb`` the matching gradient views."""
    __slots__ = ("w", "b", "gw", "gb")

    def __init__(self, w, b, gw, gb):
        self.w, self.b, self.gw, self.gb = w, b, gw, gb


def _new(shape, like: Tensor, dtype=torch.float32) -> Tensor:
    return torch.empty(shape, dtype=dtype, device=like.device)


def _zeros(shape, like: Tensor, dtype=torch.float32) -> Tensor:
    return torch.zeros(shape, dtype=dtype, device=like.device)


def _c2d(t: Tensor, rows: int, cols: int) -> Tensor:
    t = t.reshape(rows, cols)
    return t if t.stride(1) == 1 and t.stride(0) >= cols else t.contiguous()


# ---------------------------------------------------------------- encoder
class EncoderStreamFn(Function):
    """Linear(2048->D) -> HighWay -> Dropout for one modality
    (models/Encoder.py:9-25,62-66)."""

    @staticmethod
    def forward(ctx, x, cfg, *params):
        B, F, Din = x.shape
        p0: Pack = cfg["lin"]
        p12: Pack = cfg["hw"]
        D = p0.w.shape[0]
        x2 = _c2d(x, B * F, Din)
        h = _new((B * F, D), x)
        ops.linear_fwd(x2, p0.w, h, ops.Epi(bias=p0.b))
        tg = _new((B * F, 2 * D), x)
        ops.linear_fwd(h, p12.w, tg, ops.Epi(bias=p12.b, act=L.ACT_TANH_SIGMOID, act_split=D))
        out = _new((B * F, D), x)
        p = cfg["p"] if cfg["training"] else 0.0
        ops.highway_mix_fwd(h, tg, out, p, cfg["salt"], cfg["rng"])
        ctx.cfg, ctx.p = cfg, p
        ctx.x2, ctx.h, ctx.tg = x2, h, tg
        ctx.shape = (B, F, Din)
        return out.view(B, F, D)

    @staticmethod
    def backward(ctx, dout):
        cfg = ctx.cfg
        p0: Pack = cfg["lin"]
        p12: Pack = cfg["hw"]
        B, F, Din = ctx.shape
        D = p0.w.shape[0]
        d2 = _c2d(dout, B * F, D)
        dh = _new((B * F, D), d2)
        dp = _new((B * F, 2 * D), d2)
        ops.highway_mix_bwd(d2, ctx.h, ctx.tg, dh, dp, ctx.p, cfg["salt"], cfg["rng"])
        ops.linear_bwd_data(dp, p12.w, dh, beta=1.0)
        ops.linear_bwd_weight(dp, ctx.h, p12.gw, p12.gb, beta=1.0)
        ops.linear_bwd_weight(dh, ctx.x2, p0.gw, p0.gb, beta=1.0)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _new((B * F, Din), d2)
            ops.linear_bwd_data(dh, p0.w, dx)
            dx = dx.view(B, F, Din)
        ctx.x2 = ctx.h = ctx.tg = None
        return (dx, None) + (None,) * (len(ctx.needs_input_grad) - 2)


class EncoderStreamsFn(Function):
    """All modalities of the visual encoder in ONE autograd node: the same kernels as EncoderStreamFn per modality
    (models/Encoder.py:9-25,47-66), but layer by layer across the modalities, so that the streams' independent GEMMs of
    a layer can share a launch (ops.wide_group: at 128 videos one stream's Linear(2048 -> 512) is 120 big output tiles,
    half of the chip).  Results per stream are bit-identical to EncoderStreamFn.

    apply(cfgs, n_mod, x_0 .. x_{n-1}, *params of stream 0, *params of stream 1, ...) -> out_0 .. out_{n-1}"""

    @staticmethod
    def forward(ctx, cfgs, n_mod, *args):
        xs = args[:n_mod]
        st = []
        for x, cfg in zip(xs, cfgs):
            B, F, Din = x.shape
            D = cfg["lin"].w.shape[0]
            st.append(dict(shape=(B, F, Din), D=D, x2=_c2d(x, B * F, Din), h=_new((B * F, D), x), tg=_new((B * F, 2 * D), x),
                           out=_new((B * F, D), x), p=cfg["p"] if cfg["training"] else 0.0))
        with ops.wide_group():
            for s_, cfg in zip(st, cfgs):
                ops.linear_fwd(s_["x2"], cfg["lin"].w, s_["h"], ops.Epi(bias=cfg["lin"].b))
        with ops.wide_group():
            for s_, cfg in zip(st, cfgs):
                ops.linear_fwd(s_["h"], cfg["hw"].w, s_["tg"], ops.Epi(bias=cfg["hw"].b, act=L.ACT_TANH_SIGMOID, act_split=s_["D"]))
        for s_, cfg in zip(st, cfgs):
            ops.highway_mix_fwd(s_["h"], s_["tg"], s_["out"], s_["p"], cfg["salt"], cfg["rng"])
        ctx.cfgs, ctx.st, ctx.n_mod = cfgs, st, n_mod
        ctx.n_params = [len(cfg["params"]) for cfg in cfgs]
        outs = tuple(s_["out"].view(s_["shape"][0], s_["shape"][1], s_["D"]) for s_ in st)
        for s_ in st:
            del s_["out"]
        return outs

    @staticmethod
    def backward(ctx, *douts):
        cfgs, st, n_mod = ctx.cfgs, ctx.st, ctx.n_mod
        for s_, dout in zip(st, douts):
            B, F, Din = s_["shape"]
            D = s_["D"]
            s_["d2"] = _c2d(dout, B * F, D)
            s_["dh"] = _new((B * F, D), s_["d2"])
            s_["dp"] = _new((B * F, 2 * D), s_["d2"])
        for s_, cfg in zip(st, cfgs):
            ops.highway_mix_bwd(s_["d2"], s_["h"], s_["tg"], s_["dh"], s_["dp"], s_["p"], cfg["salt"], cfg["rng"])
        with ops.wide_group():
            for s_, cfg in zip(st, cfgs):
                ops.linear_bwd_data(s_["dp"], cfg["hw"].w, s_["dh"], beta=1.0)
        dxs = []
        for i, (s_, cfg) in enumerate(zip(st, cfgs)):
            ops.linear_bwd_weight(s_["dp"], s_["h"], cfg["hw"].gw, cfg["hw"].gb, beta=1.0)
            ops.linear_bwd_weight(s_["dh"], s_["x2"], cfg["lin"].gw, cfg["lin"].gb, beta=1.0)
            dx = None
            if ctx.needs_input_grad[2 + i]:
                B, F, Din = s_["shape"]
                dx = _new((B * F, Din), s_["d2"])
                ops.linear_bwd_data(s_["dh"], cfg["lin"].w, dx)
                dx = dx.view(B, F, Din)
            dxs.append(dx)
        ctx.st = None
        return (None, None) + tuple(dxs) + (None,) * sum(ctx.n_params)


_SYNC_BN_ONE_EXCHANGE = os.environ.get("NACF_SYNC_BN_EXCHANGES", "1") != "2"     # 2 = the two all-reduces of round 2
_BN_MULTI = os.environ.get("NACF_BN_MULTI", "1") != "0"      # tuning: 0 = one modality per launch (round 2)


class BNConcatFn(Function):
    """per-modality BatchNorm1d over B*F rows + temporal concat
    (models/joint_representation.py:40-51).

    cfg['sync'] (data-parallel training with opt['sync_bn']): an object with `.all_reduce(tensor)` (sum over ranks, in
    stream order) and `.world`.  The statistics are then those of the GLOBAL batch -- the reference is one process
    with the whole batch -- formed by the same two passes, each rank contributing its rows and the ranks exchanging
    one [n_modalities, D] vector per pass (two in forward, one [n_modalities, 2, D] in backward)."""

    @staticmethod
    def forward(ctx, cfg, n_mod, *args):
        xs = [a.contiguous() for a in args[:n_mod]]
        B, _, D = xs[0].shape
        M_total = sum(x.shape[1] for x in xs)
        out = _new((B, M_total, D), xs[0])
        training = cfg["training"]
        sync = cfg.get("sync") if training else None
        mods = cfg["mods"]
        mom, eps = cfg.get("momentum", 0.1), cfg.get("eps", 1e-5)
        multi = 1 <= n_mod <= 4 and _BN_MULTI        # every modality in the same launches (nacf_bn_*_multi)
        f_offs = [sum(x.shape[1] for x in xs[:i]) for i in range(n_mod)]
        sms = [_new((D,), x) if training else None for x in xs]
        sis = [_new((D,), x) if training else None for x in xs]
        stats = n_tot = None
        if sync is not None:
            stats = _new((2, n_mod, D), xs[0])
            n_tot = [x.shape[0] * x.shape[1] * sync.world for x in xs]      # every rank holds the same number of rows:
            if hasattr(sync, "assert_equal_rows"):                          # ... a descriptive error on the first calls and periodically
                sync.assert_equal_rows(xs[0].shape[0] * xs[0].shape[1])     #     (runtime/ddp.py); EVERY step is guarded on the device, below
            if getattr(sync, "all_gather", None) is not None and _SYNC_BN_ONE_EXCHANGE:
                # one exchange: (sum | squared deviations about the rank's OWN mean | this rank's row counts) gathered, merged
                # exactly.  The counts ride in the same vector: a ragged global batch makes the merged statistics NaN on every
                # rank in that very step (nacf_bn_sync_merge) -- no extra collective, no host read
                n_loc = [x.shape[0] * x.shape[1] for x in xs]
                C2 = 2 * n_mod * D
                key = (tuple(n_loc), D, str(xs[0].device))
                cache = sync.__dict__.setdefault("_bn_loc_cache", {})      # (the cfg dict is rebuilt every forward; `sync` lives with the model)
                ext = cache.get(key)
                if ext is None:          # persistent: the count tail is written once, the statistics every step (hipGraph-safe)
                    ext = torch.zeros(C2 + n_mod, dtype=xs[0].dtype, device=xs[0].device)
                    ext[C2:] = torch.tensor([float(v) for v in n_loc], dtype=xs[0].dtype)
                    cache[key] = ext
                loc = ext[:C2].view(2, n_mod, D)
                if multi:
                    ops.bn_sync_local_multi(xs, loc)
                else:
                    for i, x in enumerate(xs):
                        ops.bn_sync_stat(x, None, n_loc[i], loc[0, i])
                    for i, x in enumerate(xs):
                        ops.bn_sync_stat(x, loc[0, i], n_loc[i], loc[1, i])
                gathered = _new((sync.world, C2 + n_mod), xs[0])
                sync.all_gather(gathered, ext)
                ops.bn_sync_merge(gathered, n_loc, stats, getattr(sync, "ragged_flag", lambda d: None)(xs[0].device))
            else:
                for i, x in enumerate(xs):
                    ops.bn_sync_stat(x, None, n_tot[i], stats[0, i])
                sync.all_reduce(stats[0])
                for i, x in enumerate(xs):
                    ops.bn_sync_stat(x, stats[0, i], n_tot[i], stats[1, i])
                sync.all_reduce(stats[1])
        if multi:
            ops.bn_concat_fwd_multi(xs, out, f_offs, [m["pack"].w for m in mods], [m["pack"].b for m in mods],
                                    [m["running_mean"] for m in mods], [m["running_var"] for m in mods], [m["nbt"] for m in mods],
                                    sms, sis, training, mom, eps, stats_global=stats, n_total=n_tot)
        else:
            for i, x in enumerate(xs):
                m = mods[i]
                if sync is not None:
                    ops.bn_concat_fwd_sync(x, out, f_offs[i], m["pack"].w, m["pack"].b, m["running_mean"], m["running_var"],
                                           m["nbt"], sms[i], sis[i], stats[0, i], stats[1, i], n_tot[i], mom, eps)
                else:
                    ops.bn_concat_fwd(x, out, f_offs[i], m["pack"].w, m["pack"].b, m["running_mean"], m["running_var"],
                                      m["nbt"], sms[i], sis[i], training, mom, eps)
        ctx.cfg, ctx.n_mod, ctx.sync, ctx.multi = cfg, n_mod, sync, multi
        ctx.saves = [(x, f_offs[i], sms[i], sis[i]) for i, x in enumerate(xs)]
        return out

    @staticmethod
    def backward(ctx, dout):
        cfg = ctx.cfg
        if not cfg["training"]:
            raise L.NacfLibraryError("BNConcatFn.backward in eval mode is not supported")
        dout = dout.contiguous()
        sync, n_mod = ctx.sync, ctx.n_mod
        xs, f_offs = [s_[0] for s_ in ctx.saves], [s_[1] for s_ in ctx.saves]
        sms, sis = [s_[2] for s_ in ctx.saves], [s_[3] for s_ in ctx.saves]
        pks: List[Pack] = [cfg["mods"][i]["pack"] for i in range(n_mod)]
        dxs = [torch.empty_like(x) for x in xs]
        sums = n_tot = None
        if sync is not None:
            sums = _new((n_mod, 2, xs[0].shape[2]), dout)
            n_tot = [x.shape[0] * x.shape[1] * sync.world for x in xs]
            if ctx.multi:
                ops.bn_sync_bwd_local_multi(dout, xs, f_offs, sms, sis, sums, [pk.gw for pk in pks], [pk.gb for pk in pks], beta=1.0)
            else:
                for i, x in enumerate(xs):
                    ops.bn_sync_bwd_stat(dout, x, f_offs[i], sms[i], sis[i], sums[i], pks[i].gw, pks[i].gb, beta=1.0)     # local dW / db
            sync.all_reduce(sums)
        if ctx.multi:
            ops.bn_concat_bwd_multi(dout, xs, dxs, f_offs, [pk.w for pk in pks], sms, sis, [pk.gw for pk in pks],
                                    [pk.gb for pk in pks], beta=1.0, sums_global=sums, n_total=n_tot)
        else:
            for i, x in enumerate(xs):
                if sync is not None:
                    ops.bn_concat_bwd_sync(dout, x, dxs[i], f_offs[i], pks[i].w, sms[i], sis[i], sums[i], n_tot[i])
                else:
                    ops.bn_concat_bwd(dout, x, dxs[i], f_offs[i], pks[i].w, sms[i], sis[i], pks[i].gw, pks[i].gb, beta=1.0)
        grads = [dx if ctx.needs_input_grad[2 + i] else None for i, dx in enumerate(dxs)]
        ctx.saves = None
        return (None, None) + tuple(grads) + (None,) * (len(ctx.needs_input_grad) - 2 - n_mod)


class MeanTimeFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, T, D = x.shape
        ctx.shape = (B, T, D)
        return ops.mean_time_fwd(x, _new((B, D), x))

    @staticmethod
    def backward(ctx, dout):
        dx = _new(ctx.shape, dout)
        ops.mean_time_bwd(dout.contiguous(), dx, accumulate=False)
        return dx


class MemoryFanoutFn(Function):
    """(memory, pooled) = (x, mean_t x) as ONE autograd node.  The visual memory feeds the decoder's cross-attention (whose
    K|V projection returns a dense [B, T, D] gradient) and, through its time mean, the length head and the decoder's
    input enhancement (models/Predictor.py:29, models/Decoder.py:137).  As two nodes autograd materialises the mean's
    gradient as a second dense [B, T, D] tensor and adds the two (a 31 MB write + a 94 MB add per step at B = 128); here
    the mean's gradient is accumulated into the projection's in one read-modify-write."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, T, D = x.shape
        ctx.shape = (B, T, D)
        pooled = ops.mean_time_fwd(x, _new((B, D), x))
        # the mean as TWO outputs over one buffer, one per consumer (length head | decoder input): autograd then hands their
        # gradients over separately and the add happens inside nacf_mean_time_bwd, not in a kernel of its own
        return x.view_as(x), pooled, pooled.view_as(pooled)

    @staticmethod
    def backward(ctx, dmem, dpooled, dpooled2=None):
        if dpooled is None:
            dpooled, dpooled2 = dpooled2, None
        if dpooled is None:
            return dmem
        dpooled = dpooled.contiguous()
        dpooled2 = dpooled2.contiguous() if dpooled2 is not None else None
        if dmem is None:
            dx = _new(ctx.shape, dpooled)
            ops.mean_time_bwd(dpooled, dx, accumulate=False, dout2=dpooled2)
            return dx
        dx = dmem.contiguous()
        ops.mean_time_bwd(dpooled, dx, accumulate=True, dout2=dpooled2)      # in place: dmem is this node's alone
        return dx


class LengthHeadFn(Function):
    """Predictor_length, models/Predictor.py:12-30."""

    @staticmethod
    def forward(ctx, pooled, cfg, *params):
        p1: Pack = cfg["l1"]
        p2: Pack = cfg["l2"]
        B, D = pooled.shape
        pooled = pooled.contiguous()
        pre = _new((B, p1.w.shape[0]), pooled)
        h = _new((B, p1.w.shape[0]), pooled)
        p = cfg["p"] if cfg["training"] else 0.0
        epi = ops.Epi(bias=p1.b, act=L.ACT_RELU, preact=pre, p1=p, salt1=cfg["salt"], rng=cfg["rng"])
        ops.linear_fwd(pooled, p1.w, h, epi)
        z = _new((B, p2.w.shape[0]), pooled)
        ops.linear_fwd(h, p2.w, z, ops.Epi(bias=p2.b))
        ops.log_softmax_rows(z, z)
        ctx.cfg, ctx.epi = cfg, epi
        ctx.pooled, ctx.h, ctx.logp = pooled, h, z
        return z

    @staticmethod
    def backward(ctx, dlogp):
        cfg = ctx.cfg
        p1: Pack = cfg["l1"]
        p2: Pack = cfg["l2"]
        with ops.aux_workspace():       # may run on a side stream next to the decoder's backward (models/seq2seq.py)
            dz = torch.empty_like(ctx.logp)
            ops.log_softmax_rows_bwd(dlogp.contiguous(), ctx.logp, dz)
            dh = torch.empty_like(ctx.h)
            ops.linear_bwd_data(dz, p2.w, dh)
            ops.linear_bwd_weight(dz, ctx.h, p2.gw, p2.gb, beta=1.0)
            ops.epilogue_bwd(dh, dh, None, ctx.epi)
            dpooled = torch.empty_like(ctx.pooled)
            ops.linear_bwd_data(dh, p1.w, dpooled)
            ops.linear_bwd_weight(dh, ctx.pooled, p1.gw, p1.gb, beta=1.0)
        return (dpooled, None) + (None,) * (len(ctx.needs_input_grad) - 2)


# ---------------------------------------------------------------- decoder
def _embed_ln_forward(ctx, additional, cfg, tokens, category, word_w):
    R, Lq = tokens.shape
    pos, cat = cfg["pos"], cfg["cat"]
    D = word_w.shape[1]
    training = cfg["training"]
    out = _new((R, Lq, D), word_w)
    xhat = _new((R * Lq, D), word_w) if training else None
    rstd = _new((R * Lq,), word_w) if training else None
    p = cfg["p"] if training else 0.0
    add = additional.contiguous() if additional is not None else None
    ops.embed_ln_fwd(tokens, category if cat is not None else None, add, word_w, pos.w,
                     cat.w if cat is not None else None, cfg["ln"].w, cfg["ln"].b, out, xhat, rstd,
                     cfg["vdiv"], cfg["vmod"], cfg["eps"], p, cfg["salt"], cfg["rng"])
    ctx.cfg, ctx.p = cfg, p
    ctx.tokens, ctx.category, ctx.xhat, ctx.rstd = tokens, category, xhat, rstd
    ctx.n_video = additional.shape[0] if additional is not None else 0
    return out


def _embed_ln_backward(ctx, dout, V, D, word_gw, want_dadd):
    cfg = ctx.cfg
    pos, cat, ln = cfg["pos"], cfg["cat"], cfg["ln"]
    R, Lq = ctx.tokens.shape
    dE = _new((R * Lq, D), dout)
    ops.embed_ln_bwd(dout.contiguous(), ctx.xhat, ctx.rstd, ln.w, dE, ln.gw, ln.gb, R, Lq, D, ctx.p,
                     cfg["salt"], cfg["rng"], beta=1.0)
    dadd = _new((ctx.n_video, D), dout) if (ctx.n_video and want_dadd) else None
    ops.embed_scatter_bwd(dE, ctx.tokens, ctx.category if cat is not None else None, word_gw, pos.gw,
                          cat.gw if cat is not None else None, dadd, R, Lq, D, V,
                          cat.w.shape[0] if cat is not None else 0, cfg["vmod"], cfg["vdiv"], cfg["vmod"])
    ctx.xhat = ctx.rstd = None
    return dadd


class EmbedLNFn(Function):
    """BertEmbeddings, models/bert.py:70-96."""

    @staticmethod
    def forward(ctx, additional, cfg, tokens, category, *params):
        return _embed_ln_forward(ctx, additional, cfg, tokens, category, cfg["word"].w)

    @staticmethod
    def backward(ctx, dout):
        word = ctx.cfg["word"]
        dadd = _embed_ln_backward(ctx, dout, word.w.shape[0], word.w.shape[1],
                                  word.gw if ctx.cfg.get("train_word", True) else None, ctx.needs_input_grad[0])
        return (dadd, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 4)


class EmbedLNTableFn(Function):
    """BertEmbeddings over a word table that is itself a computed tensor: opt['load_word_embeddings'] (models/bert.py:51-53,
    77-79) looks 768-wide rows up and projects them to dim_hidden, i.e. it looks up rows of
    `table = word_embeddings.weight @ prj.weight^T + prj.bias` ([V, D], one GEMM over the vocabulary instead of one over
    every token; the <pad> row of the 768-wide table is zero, so table[<pad>] = prj.bias as upstream).  The gradient of the
    table ([V, D]: the deterministic scatter of models' dE, <pad> row untouched = 0, as nn.Embedding's padding_idx leaves
    it) goes back through autograd to the projection's LinearFn."""

    @staticmethod
    def forward(ctx, additional, table, cfg, tokens, category, *params):
        ctx.table_shape = tuple(table.shape)
        return _embed_ln_forward(ctx, additional, cfg, tokens, category, table.contiguous())

    @staticmethod
    def backward(ctx, dout):
        V, D = ctx.table_shape
        dtable = torch.zeros(V, D, dtype=dout.dtype, device=dout.device) if ctx.needs_input_grad[1] else None
        dadd = _embed_ln_backward(ctx, dout, V, D, dtable, ctx.needs_input_grad[0])
        return (dadd, dtable, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 5)


class ProjectTableFn(Function):
    """table = word_embeddings.weight @ prj.weight^T + prj.bias  ([V, D]; opt['load_word_embeddings'], models/bert.py:51-53,
    77-79).  Like every other node of this runtime it writes its parameter gradients straight into the flat gradient buffer
    (cfg['table'].gw, cfg['pack'].gw / .gb) instead of handing them to autograd's accumulation nodes, which the captured
    training step does not go through."""

    @staticmethod
    def forward(ctx, cfg, *params):
        table, prj = cfg["table"], cfg["pack"]
        out = _new((table.w.shape[0], prj.w.shape[0]), table.w)
        ops.linear_fwd(table.w, prj.w, out, ops.Epi(bias=prj.b))
        ctx.cfg = cfg
        return out

    @staticmethod
    def backward(ctx, dtable):
        table, prj = ctx.cfg["table"], ctx.cfg["pack"]
        dtable = dtable.contiguous()
        if table.gw is not None and ctx.cfg.get("train_word", True):
            ops.linear_bwd_data(dtable, prj.w, table.gw, beta=1.0)          # <pad> row of dtable is zero: so is its gradient row
        if prj.gw is not None:
            ops.linear_bwd_weight(dtable, table.w, prj.gw, prj.gb, beta=1.0)
        return (None,) * len(ctx.needs_input_grad)


class PositionRowsFn(Function):
    """position_embeddings.weight[:Lq] repeated for R sequences: [R * Lq, D] (the input of pos_LN, models/bert.py:97-105).
    Backward sums the R copies into the table's rows of the flat gradient buffer (the position part of the embedding
    scatter: fixed summation order), not through autograd's accumulation nodes (see ProjectTableFn)."""

    @staticmethod
    def forward(ctx, cfg, tokens, *params):
        R, Lq = tokens.shape
        ctx.cfg, ctx.tokens = cfg, tokens
        return cfg["pos"].w[:Lq].repeat(R, 1)

    @staticmethod
    def backward(ctx, drows):
        pos = ctx.cfg["pos"]
        R, Lq = ctx.tokens.shape
        D = pos.w.shape[1]
        if pos.gw is not None:
            ops.embed_scatter_bwd(drows.contiguous(), ctx.tokens, None, None, pos.gw, None, None, R, Lq, D, 0, 0, 0, 1, 1)
        return (None,) * len(ctx.needs_input_grad)


# NACF_DEAD_ROWS=fill (tuning / A-B): round 3's policy -- every row-list GEMM zero-fills its dead rows, the epilogue backward
# walks every row.  Default: only where something reads every row (models/bert.py:BertLayer.run).
_FILL_ALL = os.environ.get("NACF_DEAD_ROWS", "") == "fill"


class LinearFn(Function):
    """nn.Linear with the fused epilogue of nacf_linear_fwd.
    cfg keys: pack, act, p1, salt1, p2, salt2, row_tokens, rng, training."""

    @staticmethod
    def forward(ctx, x, residual, cfg, *params):
        pk: Pack = cfg["pack"]
        M, K = x.shape
        N = pk.w.shape[0]
        x = _c2d(x, M, K)
        training = cfg.get("training", False)
        act = cfg.get("act", L.ACT_NONE)
        need_bwd = any(ctx.needs_input_grad)
        rows = cfg.get("rows")      # live-row list: dead (<pad>) slots are not computed; the GEMM zero-fills them
        mk = _new
        pre = mk((M, N), x) if (act != L.ACT_NONE and need_bwd) else None
        res = _c2d(residual, M, N) if residual is not None else None
        epi = ops.Epi(bias=pk.b, act=act, preact=pre,
                      p1=cfg.get("p1", 0.0) if training else 0.0, salt1=cfg.get("salt1", 0),
                      residual=res, p2=cfg.get("p2", 0.0) if training else 0.0, salt2=cfg.get("salt2", 0),
                      row_tokens=cfg.get("row_tokens"), rng=cfg.get("rng"))
        out = mk((M, N), x)
        # cfg['fill'] = False: every consumer of `out` (and of the pre-activation) walks the same row list, so the dead rows
        # need no zeros (a 2048-wide FFN1 output of 5120 slots: 36 MB of zero stores per step with its pre-activation)
        ops.linear_fwd(x, pk.w, out, epi, rows, zero_dead=bool(cfg.get("fill", True)) or _FILL_ALL)
        ctx.cfg, ctx.epi, ctx.x, ctx.rows = cfg, epi, x, rows
        ctx.has_res = residual is not None
        return out

    @staticmethod
    def backward(ctx, dy):
        cfg, epi = ctx.cfg, ctx.epi
        pk: Pack = cfg["pack"]
        M, K = ctx.x.shape
        N = pk.w.shape[0]
        dy = _c2d(dy, M, N)
        trivial = (epi.act == L.ACT_NONE and epi.p1 == 0.0 and epi.p2 == 0.0 and epi.row_tokens is None
                   and not ctx.has_res)
        dr = None
        if trivial:
            dz = dy
        else:
            dz = _new((M, N), dy)
            dr = _new((M, N), dy) if ctx.has_res else None
            epi.residual = dr  # only its presence matters to the kernel
            # (live rows only; dead rows of dr: zeros, of dz: untouched)
            ops.epilogue_bwd(dy, dz, dr, epi, rows=None if _FILL_ALL else ctx.rows)
            epi.residual = None
        dx = None
        rows = ctx.rows
        # residual-gradient hand-over (cfg['res_sink'] / cfg['dx_acc'], see BertLayer.run): when x feeds both this
        # Linear and the residual input of a LATER Linear, that one parks its residual gradient in the shared holder
        # and this dX GEMM accumulates onto it (beta = 1): no separate gradient-add kernel, x gets the sum.
        sink = cfg.get("res_sink")
        if sink is not None and dr is not None and ctx.needs_input_grad[1]:
            sink["dr"] = dr
            dr = None
        acc = cfg.get("dx_acc")
        if ctx.needs_input_grad[0]:
            parked = acc.pop("dr", None) if acc is not None else None
            if parked is not None:
                dx = parked
                ops.linear_bwd_data(dz, pk.w, dx, beta=1.0, rows=rows)
            else:
                dx = _new((M, K), dy)
                # cfg['dx_fill'] = False: the producer of x backpropagates through a row-list epilogue (LinearFn of the same list)
                ops.linear_bwd_data(dz, pk.w, dx, rows=rows, zero_dead=bool(cfg.get("dx_fill", True)) or _FILL_ALL)
        if pk.gw is not None:
            ops.linear_bwd_weight(dz, ctx.x, pk.gw, pk.gb, beta=1.0, rows=rows)
        ctx.x = None
        epi.preact = None
        return (dx, dr if ctx.needs_input_grad[1] else None, None) + (None,) * (len(ctx.needs_input_grad) - 3)


class HalvesFn(Function):
    """x[2B, ...] -> (x[:B], x[B:]) (the two decoding passes that were batched into one set of rows).  Backward
    writes both gradients into ONE buffer; plain slicing costs two zero-fills, two copies and an add."""

    @staticmethod
    def forward(ctx, x):
        B = x.shape[0] // 2
        ctx.shape = x.shape
        return x[:B], x[B:]

    @staticmethod
    def backward(ctx, g0, g1):
        B = ctx.shape[0] // 2
        g = torch.empty(ctx.shape, dtype=(g0 if g0 is not None else g1).dtype, device=(g0 if g0 is not None else g1).device)
        if g0 is not None: g[:B].copy_(g0)
        else: g[:B].zero_()
        if g1 is not None: g[B:].copy_(g1)
        else: g[B:].zero_()
        return g


class SelfAttentionFn(Function):
    """softmax(QK^T/sqrt(dk) masked) V on a packed [rows, 3D] q|k|v buffer
    (models/bert.py:150-179; masks models/Decoder.py:13-39,105-124)."""

    @staticmethod
    def forward(ctx, qkv, tokens, causal, H, want_probs, drop=None):
        R, Lq = tokens.shape
        D = qkv.shape[1] // 3
        dk = D // H
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        out = _new((R * Lq, D), qkv)
        probs = _new((H, R, Lq, Lq), qkv) if want_probs else None
        ops.attention_fwd(q, k, v, out, tokens, causal, probs, R, H, Lq, Lq, dk, 1, R, drop=drop)
        ctx.qkv, ctx.tokens, ctx.causal, ctx.H, ctx.drop = qkv, tokens, causal, H, drop
        if want_probs:
            ctx.mark_non_differentiable(probs)
            return out, probs
        return out, None

    @staticmethod
    def backward(ctx, do, _dprobs=None):
        qkv, tokens = ctx.qkv, ctx.tokens
        R, Lq = tokens.shape
        D = qkv.shape[1] // 3
        dk = D // ctx.H
        do = _c2d(do, R * Lq, D)
        dqkv = torch.empty_like(qkv)
        ops.attention_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], do, dqkv[:, :D], dqkv[:, D:2 * D],
                          dqkv[:, 2 * D:], tokens, ctx.causal, R, R, ctx.H, Lq, Lq, dk, 1, R, drop=ctx.drop)
        ctx.qkv = None
        return dqkv, None, None, None, None, None


class QKVAttentionFn(Function):
    """self-attention geometry with q|k taken from one packed [rows, 2D] tensor and v from another [rows, D] tensor
    (the position attention of models/bert.py:274-281: queries and keys from the position embeddings, values from
    the hidden states)."""

    @staticmethod
    def forward(ctx, qk, v, tokens, causal, H, drop=None):
        R, Lq = tokens.shape
        D = v.shape[1]
        dk = D // H
        out = _new((R * Lq, D), v)
        ops.attention_fwd(qk[:, :D], qk[:, D:], v, out, tokens, causal, None, R, H, Lq, Lq, dk, 1, R, drop=drop)
        ctx.qk, ctx.v, ctx.tokens, ctx.causal, ctx.H, ctx.drop = qk, v, tokens, causal, H, drop
        return out

    @staticmethod
    def backward(ctx, do):
        qk, v, tokens = ctx.qk, ctx.v, ctx.tokens
        R, Lq = tokens.shape
        D = v.shape[1]
        dk = D // ctx.H
        do = _c2d(do, R * Lq, D)
        dqk, dv = torch.empty_like(qk), torch.empty_like(v)
        ops.attention_bwd(qk[:, :D], qk[:, D:], v, do, dqk[:, :D], dqk[:, D:], dv, tokens, ctx.causal, R, R, ctx.H, Lq, Lq,
                          dk, 1, R, drop=ctx.drop)
        ctx.qk = ctx.v = None
        return dqk, dv, None, None, None, None


class CrossAttentionFn(Function):
    """decoder -> visual memory attention; K|V are the packed projection of the
    memory, computed once per video and shared by every row mapped to it."""

    @staticmethod
    def forward(ctx, q, kv, H, Lq, Lk, kv_div, kv_mod, want_probs, drop=None):
        D = q.shape[1]
        dk = D // H
        R = q.shape[0] // Lq
        k, v = kv[:, :D], kv[:, D:]
        out = _new((R * Lq, D), q)
        probs = _new((H, R, Lq, Lk), q) if want_probs else None
        if kv_div > 1 and R == kv_div * kv_mod and not want_probs and not any(ctx.needs_input_grad) and drop is None:
            # inference with the length beam: the kv_div candidates of a video are consecutive rows and nothing masks a
            # query, so they are ONE sequence of kv_div*Lq queries over that video's memory -- 114 rows in blocks of
            # 32 instead of 6 x (19 padded to 32), and the blocks share the K / V rows they stream
            ops.attention_fwd(q, k, v, out, None, 0, None, kv_mod, H, Lq * kv_div, Lk, dk, 1, kv_mod)
        else:
            ops.attention_fwd(q, k, v, out, None, 0, probs, R, H, Lq, Lk, dk, kv_div, kv_mod, drop=drop)
        ctx.q, ctx.kv, ctx.drop = q, kv, drop
        ctx.dims = (R, H, Lq, Lk, dk, kv_div, kv_mod)
        if want_probs:
            ctx.mark_non_differentiable(probs)
            return out, probs
        return out, None

    @staticmethod
    def backward(ctx, do, _dprobs=None):
        R, H, Lq, Lk, dk, kv_div, kv_mod = ctx.dims
        q, kv = ctx.q, ctx.kv
        D = q.shape[1]
        n_kv = kv.shape[0] // Lk
        do = _c2d(do, R * Lq, D)
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        ops.attention_bwd(q, kv[:, :D], kv[:, D:], do, dq, dkv[:, :D], dkv[:, D:], None, 0, R, n_kv, H, Lq, Lk, dk,
                          kv_div, kv_mod, drop=ctx.drop)
        ctx.q = ctx.kv = None
        return dq, dkv, None, None, None, None, None, None, None


# ---------------------------------------------------------------- vocabulary
class VocabLogProbFn(Function):
    """tgt_word_prj + log_softmax (models/seq2seq.py:102-103); returns real
    [rows, V] log-probs so any criterion (incl. the reference's) can consume them."""

    @staticmethod
    def forward(ctx, h, cfg, *params):
        pk: Pack = cfg["pack"]
        rows, D = h.shape
        V = pk.w.shape[0]
        h = _c2d(h, rows, D)
        buf = _new((rows, ops.vocab_ld(V)), h)
        logits = buf[:, :V]
        ops.linear_fwd(h, pk.w, logits, ops.Epi(bias=pk.b))
        ops.vocab_logsoftmax_fwd(logits, V, None, None, None, None)
        ctx.cfg, ctx.h, ctx.logp = cfg, h, logits
        return logits

    @staticmethod
    def backward(ctx, dlogp):
        pk: Pack = ctx.cfg["pack"]
        rows, V = ctx.logp.shape
        if dlogp.stride(1) != 1:
            dlogp = dlogp.contiguous()
        dbuf = _new((rows, ops.vocab_ld(V)), dlogp)
        if ops.vocab_ld(V) != V:
            dbuf[:, V:].zero_()
        dlogits = dbuf[:, :V]
        ops.vocab_logsoftmax_bwd(dlogp, ctx.logp, dlogits, V)
        dh = torch.empty_like(ctx.h)
        ops.linear_bwd_data(dlogits, pk.w, dh)
        ops.linear_bwd_weight(dlogits, ctx.h, pk.gw, pk.gb, beta=1.0)
        ctx.h = ctx.logp = None
        return (dh, None) + (None,) * (len(ctx.needs_input_grad) - 2)


class FusedVocabXentFn(Function):
    """Vocabulary projection + log-softmax + NLL in one Function (K9): returns
    stats[5] = (token-sum NLL, acc hits, acc count, sum logp, token count)
    (misc/crit.py:62-114); the [rows, V] buffer is reused for the gradient."""

    @staticmethod
    def forward(ctx, h, cfg, labels, exclude_mask, *params):
        pk: Pack = cfg["pack"]
        rows, D = h.shape
        V = pk.w.shape[0]
        h = _c2d(h, rows, D)
        labels = labels.reshape(-1).contiguous()
        # only slots that carry a label (label != PAD) reach the loss, the meters or the gradient:
        # project / normalise / back-propagate just those rows (live-row list built from the labels)
        live = ops.rowset_build(tokens=labels)
        buf = _new((rows, ops.vocab_ld(V)), h)
        logits = buf[:, :V]
        label_logp = _new((rows,), h)
        lse = _new((rows,), h)
        argmax = _new((rows,), h, torch.int64)
        # projection with the soft-max statistics out of the GEMM epilogue: the [rows, V] logits are written once and
        # not read again until the backward pass turns them into the gradient in place
        ops.vocab_lse_fwd(h, pk.w, pk.b, logits, labels, lse, argmax, label_logp, live)
        stats = cfg.get("out")          # optional slot of the criterion's term slab (LossCombineFn)
        if stats is None:
            stats = _new((5,), h)
        tail = cfg.get("tail")          # ops.CritTail: the reduction runs in the criterion's ONE tail launch, which writes `stats`
        if tail is not None and cfg.get("out") is not None:
            tail.add_pass(label_logp, argmax, labels, exclude_mask, cfg["slot"])
        else:
            ops.nll_reduce(label_logp, argmax, labels, exclude_mask, stats)
        ctx.cfg, ctx.h, ctx.logp, ctx.labels, ctx.live, ctx.lse = cfg, h, logits, labels, live, lse
        return stats

    @staticmethod
    def backward(ctx, dstats):
        pk: Pack = ctx.cfg["pack"]
        rows, V = ctx.logp.shape
        dstats = dstats.contiguous()
        ops.xent_bwd_lse(ctx.logp, ctx.lse, ctx.logp, V, ctx.labels, dstats, 1.0, skip_pad_rows=True)  # in place: logits -> dlogits
        dh = torch.empty_like(ctx.h)
        ops.linear_bwd_data(ctx.logp, pk.w, dh, rows=ctx.live, zero_dead=True)
        ops.linear_bwd_weight(ctx.logp, ctx.h, pk.gw, pk.gb, beta=1.0, rows=ctx.live)
        ctx.h = ctx.logp = ctx.live = ctx.lse = None
        return (dh, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 4)


class BatchedPasses(list):
    """[pass0 hidden, pass1 hidden] that are the two halves of ONE [2B, L, D] tensor (`.both`): consumers that can
    work on the batch (the fused vocabulary loss) take `.both`, everything else sees a plain list."""

    def __init__(self, parts, both):
        super().__init__(parts)
        self.both = both


class FusedVocabXentMultiFn(Function):
    """FusedVocabXentFn over several passes that share tgt_word_prj and sit back to back in one [S*rows, D]
    tensor: ONE projection GEMM, one log-softmax, one dX and one dW GEMM for all of them (the two NACF passes
    are 821 + 1490 live rows at B=128: one launch has fewer ragged tiles than two).  Per-pass statistics and
    per-pass upstream gradients are kept apart by row range.  cfg: pack, outs (optional slab slots)."""

    @staticmethod
    def forward(ctx, h, cfg, labels, excludes, *params):
        pk: Pack = cfg["pack"]
        rows, D = h.shape
        S = len(excludes)
        rp = rows // S
        V = pk.w.shape[0]
        h = _c2d(h, rows, D)
        labels = labels.reshape(-1).contiguous()
        live = ops.rowset_build(tokens=labels)
        buf = _new((rows, ops.vocab_ld(V)), h)
        logits = buf[:, :V]
        label_logp = _new((rows,), h)
        lse = _new((rows,), h)
        argmax = _new((rows,), h, torch.int64)
        ops.vocab_lse_fwd(h, pk.w, pk.b, logits, labels, lse, argmax, label_logp, live)
        outs = cfg.get("outs")
        stats = [outs[i] if outs is not None else _new((5,), h) for i in range(S)]
        tail = cfg.get("tail")          # ops.CritTail: the reductions run in the criterion's ONE tail launch, which writes `stats`
        if tail is not None and outs is not None and S <= 4:
            for i in range(S):
                sl = slice(i * rp, (i + 1) * rp)
                tail.add_pass(label_logp[sl], argmax[sl], labels[sl], excludes[i], cfg["slots"][i])
        elif 1 < S <= 4:
            ops.nll_reduce_multi(label_logp, argmax, labels, excludes, stats)       # every pass in one launch
        else:
            for i in range(S):
                sl = slice(i * rp, (i + 1) * rp)
                ops.nll_reduce(label_logp[sl], argmax[sl], labels[sl], excludes[i], stats[i])
        ctx.cfg, ctx.h, ctx.logp, ctx.labels, ctx.live, ctx.S, ctx.lse = cfg, h, logits, labels, live, S, lse
        return tuple(stats)

    @staticmethod
    def backward(ctx, *dstats):
        pk: Pack = ctx.cfg["pack"]
        rows, V = ctx.logp.shape
        rp = rows // ctx.S
        gs = [g.contiguous() if g is not None else torch.zeros(5, dtype=ctx.logp.dtype, device=ctx.logp.device) for g in dstats]
        if 1 < ctx.S <= 4:
            ops.xent_bwd_lse_multi(ctx.logp, ctx.lse, ctx.logp, V, ctx.labels, gs, 1.0, skip_pad_rows=True)    # in place, one launch
        else:
            for i, g in enumerate(gs):
                sl = slice(i * rp, (i + 1) * rp)
                ops.xent_bwd_lse(ctx.logp[sl], ctx.lse[sl], ctx.logp[sl], V, ctx.labels[sl], g, 1.0, skip_pad_rows=True)
        dh = torch.empty_like(ctx.h)
        ops.linear_bwd_data(ctx.logp, pk.w, dh, rows=ctx.live, zero_dead=True)
        ops.linear_bwd_weight(ctx.logp, ctx.h, pk.gw, pk.gb, beta=1.0, rows=ctx.live)
        ctx.h = ctx.logp = ctx.live = ctx.lse = None
        return (dh, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 4)


class KLDivMeanFn(Function):
    """legacy nn.KLDivLoss() ('mean' over all elements), misc/crit.py:223."""

    @staticmethod
    def forward(ctx, x, t, out=None, tail=None, slot=0):
        x, t = x.contiguous(), t.contiguous()
        if out is None:                 # else: a 1-element slot of the criterion's term slab (LossCombineFn)
            out = _new((1,), x)
            tail = None
        if tail is not None:            # ops.CritTail: the criterion's ONE tail launch computes the term (and, backward, dx)
            tail.kl = (x, t, int(slot))
        else:
            ops.kldiv_mean(x, t, out, None)
        ctx.x, ctx.t, ctx.tail = x, t, tail
        return out.view(())

    @staticmethod
    def backward(ctx, dout):
        pad = (None,) * (len(ctx.needs_input_grad) - 1)
        if ctx.tail is not None and ctx.tail.kl_dx is not None:
            dx, ctx.tail.kl_dx = ctx.tail.kl_dx, None          # written by LossCombineFn.backward's tail launch
            return (dx,) + pad
        dx = torch.empty_like(ctx.x)
        ops.kldiv_mean(ctx.x, ctx.t, None, dx, gscale=dout.reshape(1).contiguous())
        return (dx,) + pad


class LossCombineFn(Function):
    """Criterion tail (misc/crit.py:40-45,107-121) in one launch each way: total = sum_t coef[t] * term_t[0], and the
    running meters (loss sums, accuracy hits/counts, perplexity sums) accumulate in the same kernel.  The terms were
    written by their producers straight into cfg['slab'] (slot t = slab[t*stride:(t+1)*stride])."""

    @staticmethod
    def forward(ctx, cfg, *terms):
        slab, coef = cfg["slab"], cfg["coef"]
        total = _new((1,), slab)
        tail = cfg.get("tail")          # ops.CritTail filled by the terms' producers: their reductions + this combine in ONE launch
        if tail is not None and (tail.passes or tail.kl is not None):
            ops.crit_tail_fwd(tail, slab, len(terms), cfg["stride"], coef, total, cfg.get("m_dst"), cfg.get("m_src"),
                              cfg.get("m_scale"), cfg.get("meters"))
        else:
            tail = None
            ops.loss_combine(slab, len(terms), cfg["stride"], coef, total, cfg.get("m_dst"), cfg.get("m_src"),
                             cfg.get("m_scale"), cfg.get("meters"))
        ctx.cfg, ctx.shapes, ctx.tail = cfg, [t.shape for t in terms], tail
        return total.view(())

    @staticmethod
    def backward(ctx, g):
        cfg, stride = ctx.cfg, ctx.cfg["stride"]
        n = len(ctx.shapes)
        gslab = _new((n * stride,), cfg["slab"])
        if ctx.tail is not None:
            kl = ctx.tail.kl
            ctx.tail.kl_dx = torch.empty_like(kl[0]) if kl is not None else None
            ops.crit_tail_bwd(ctx.tail, g.reshape(1).contiguous(), cfg["coef"], n, stride, gslab, ctx.tail.kl_dx)
        else:
            ops.loss_combine_bwd(g.reshape(1).contiguous(), cfg["coef"], n, stride, gslab)
        grads = []
        for t, shp in enumerate(ctx.shapes):
            k = 1
            for d in shp:
                k *= d
            grads.append(gslab[t * stride:t * stride + k].view(shp))
        return (None,) + tuple(grads)


class LayerNormFn(Function):
    """nn.LayerNorm over the last dimension with the optional tail of the blocks that use it
    (with_layernorm: BertSelfOutput / BertOutput, models/bert.py:189-200,237-247):
        out = (row is <pad> ? 0 : dropout(LN(x)))
    cfg keys: ln (Pack), eps, p, salt, rng, training, row_tokens."""

    @staticmethod
    def forward(ctx, x, cfg, *params):
        rows, D = x.shape
        x = _c2d(x, rows, D)
        ln: Pack = cfg["ln"]
        training = cfg.get("training", False)
        need = ctx.needs_input_grad[0] or ln.gw is not None
        xhat = _new((rows, D), x) if (training or need) else None
        rstd = _new((rows,), x) if (training or need) else None
        p = cfg.get("p", 0.0) if training else 0.0
        out = _new((rows, D), x)
        ops.layernorm_fwd(x, ln.w, ln.b, out, xhat, rstd, cfg["eps"], rows, rows, 0, p, cfg.get("salt", 0),
                          cfg.get("rng"), cfg.get("row_tokens"))
        ctx.cfg, ctx.p, ctx.xhat, ctx.rstd = cfg, p, xhat, rstd
        return out

    @staticmethod
    def backward(ctx, dout):
        cfg = ctx.cfg
        ln: Pack = cfg["ln"]
        rows, D = ctx.xhat.shape
        dx = _new((rows, D), dout)
        ops.layernorm_bwd(_c2d(dout, rows, D), ctx.xhat, ctx.rstd, ln.w, dx, ln.gw, ln.gb, rows, rows, 0, ctx.p,
                          cfg.get("salt", 0), cfg.get("rng"), cfg.get("row_tokens"), beta=1.0)
        ctx.xhat = ctx.rstd = None
        return (dx, None) + (None,) * (len(ctx.needs_input_grad) - 2)


class LNConcatFn(Function):
    """per-modality LayerNorm + temporal concat (norm_type='ln', models/joint_representation.py:21,47-51)"""

    @staticmethod
    def forward(ctx, cfg, n_mod, *args):
        xs = [a.contiguous() for a in args[:n_mod]]
        B, _, D = xs[0].shape
        M_total = sum(x.shape[1] for x in xs)
        out = _new((B, M_total, D), xs[0])
        saves, f_off = [], 0
        for i, x in enumerate(xs):
            F_ = x.shape[1]
            pk: Pack = cfg["packs"][i]
            xhat, rstd = _new((B * F_, D), x), _new((B * F_,), x)
            ops.layernorm_fwd(x.view(B * F_, D), pk.w, pk.b, out, xhat, rstd, cfg["eps"], F_, M_total, f_off, 0.0, 0,
                              None, None)
            saves.append((xhat, rstd, F_, f_off))
            f_off += F_
        ctx.cfg, ctx.saves, ctx.n_mod, ctx.dims = cfg, saves, n_mod, (B, M_total, D)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, M_total, D = ctx.dims
        dout = dout.contiguous()
        grads = []
        for i, (xhat, rstd, F_, f_off) in enumerate(ctx.saves):
            pk: Pack = ctx.cfg["packs"][i]
            dx = _new((B * F_, D), dout)
            ops.layernorm_bwd(dout, xhat, rstd, pk.w, dx, pk.gw, pk.gb, F_, M_total, f_off, 0.0, 0, None, None, beta=1.0)
            grads.append(dx.view(B, F_, D) if ctx.needs_input_grad[2 + i] else None)
        ctx.saves = None
        return (None, None) + tuple(grads) + (None,) * (len(ctx.needs_input_grad) - 2 - ctx.n_mod)
