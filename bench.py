#!/usr/bin/env python
"""bench.py -- the hot path of BASELINE.json on N GPUs of one node.

  python bench.py --gpus N --steps K --warmup W          (N=1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Headline workload (BASELINE.json configs[2]/[3]): NACF, MSRVTT-shape synthetic batch -- 128 videos per
GPU, motion+image features 60x2048 fp32 each (U[0,1)), seq_len 20, V = 10547, category embeddings, dropout 0.5 as in
the reference defaults, random-init weights.  One "step" = zero_grad + forward (two decoder passes) + fused loss +
backward + [RCCL all-reduce of the flat gradient buckets] + clip(+-5) + Adam, inputs resident in HBM, the whole step
one hipGraph.  value = global videos/s over EXACTLY --steps steps (barrier + synchronize on both sides, max over ranks).

Arithmetic: every GEMM runs in the library's default mode `bf16x3` -- fp32 operands split exactly into three bf16
terms, six v_mfma_f32_16x16x32_bf16 per product block, fp32 accumulate: fp32-accurate (the parity tests run this mode:
logits <= 1e-3, greedy NA tokens bit-exact), hence "dtype": "f32".  --gemm-mode f32 | bf16 selects the fp32 MFMA or the
bf16 throughput mode for the headline leg; the bf16 mode is always reported as its own leg (configs[1]).

Nothing is cached across steps; the only work not executed is work whose result is identically zero in the reference
too: decoder rows whose token is <pad> and vocabulary rows without a label (live-row GEMMs).
The rooflines count only FLOPs actually executed, as 2*M_live*N*K (never the 6x MFMA work of the split).

Same JSON line (rank 0, N = 1): `roofline` (dominant GEMM kernel, live HIP-event timing), `cpu_baseline` (the oracle
on the host cores, bounded sample: all cores and one thread), `decode` (NA mask-predict + coarse templates, with its
own roofline and CPU baseline), `train_L30` (the reference's MSRVTT default length), `config1_nab_bf16`,
`nacf_bf16`, `config5_ar_vs_na`, `loader_fed`.
"""
import argparse
import glob
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

METRIC = "training videos/sec (whole node) + NA-decode captions/sec, NACF MSRVTT-shape"
# MI355X_MICROARCH.md: dense MFMA peaks at 2.4 GHz
PEAK_F32_MFMA_TFLOPS = 157.3          # v_mfma_f32_16x16x4_f32
PEAK_BF16_MFMA_TFLOPS = 2500.0        # v_mfma_f32_16x16x32_bf16 / 32x32x16
MODE_PEAK = {"f32": PEAK_F32_MFMA_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS, "bf16x3": PEAK_BF16_MFMA_TFLOPS / 6.0}
# what keeps each kernel family below its roof (rocprofv3 --pmc, tools/bf16_trace.py, tools/probes/*: DESIGN.md section 4)
LIMITER_128 = ("LDS bandwidth: per 128x128 k-tile a workgroup reads 96 KB of bf16x3 fragments (24 ds_read_b128 per wave, 768 cycles "
               "at the 128 B/clk/CU the LDS delivers, tools/probes/compute_phase.hip) and stores 48 KB (~600 cycles) against 1536 "
               "cycles of MFMAs, shared by both resident workgroups; plus ~20 % of a K = 512 workgroup's life in prologue + "
               "epilogue: SQ_VALU_MFMA_BUSY ~55 % of active cycles (exact mode; ~20 % in the bf16 mode); `bound` names the roof "
               "the kernel is priced against, not a saturated unit")
LIMITER_G256W = ("the operand split: a k-tile of 32 reduce rows is 192 v_mfma_f32_16x16x32_bf16 per wave (3360 cycles at the measured ~17.5 per "
                 "instruction, two waves per SIMD) against ~530 vector instructions per wave that cut the fp32 LDS image into three bf16 planes at "
                 "fragment time (every wave for its own fragments) plus 4 barriers and 8 DMA requests; SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) "
                 "= 45 % over the whole launch (49-54 % in-loop), 3.1 other vector instructions per matrix instruction, no LDS bank conflict (profiles/r05_dw_g256_pmc.txt) "
                 "(204-224 TF of 416.7 on a 4096 cube, tools/probes/gemm256w_probe.hip; ablations in profiles/r05_g256w_ablation.txt); a grouped "
                 "launch adds the ragged last round of its 12 problems and the slab combine")
LIMITER_DMA128 = ("two resources at once: a 128 x 128 exact-mode k-tile moves 40 KB through the CU's L1 / LDS-DMA path (26.7 B/clk/CU at the matrix roof; "
                  "the path delivers 32-42: a DMA-only loop runs 2350-2600 cycles per k-tile and workgroup) and needs 2 x 1536 matrix cycles per pair of "
                  "workgroups; measured 3550 with the split hidden to 60 % behind the wave's own matrix instructions (tools/probes/dma128_probe.hip, "
                  "profiles/r06_dma128_probe.txt); two workgroups per CU cap the tile at 128 x 128 (80 KB LDS, 256 registers)")
LIMITER_DMA64 = ("launch latency and fp32 outputs, not the matrix pipe: a launch of <= 768 tiles of 32 x 64 is 2-12 GFLOP (1-5 us of the chip's "
                 "bf16 rate) inside a ~4.7 us launch floor, one HBM latency of prologue and an epilogue that writes fp32 (with the pre-activation: "
                 "8 bytes per output against 2 x 512 flops); per launch 7-18 us (profiles/r05_dma64_timeline.txt)")
LIMITER_WIDE = ("inside the k-loop the matrix pipe is 93 % busy (3250-3310 cycles per k-tile of 96 v_mfma_f32_32x32x16_bf16 = 3072; "
                "64-row tiles 81 %: tools/probes/wide_gemm.hip stamps); what is left is outside it: one workgroup per CU, so "
                "nothing overlaps its prologue (~4 k cycles: DMA latency + the first operand split) and epilogue (10-14 k cycles, "
                "HBM-write bound when all CUs store at once) -- 23 % of a K = 512 workgroup's life -- nor a partly filled last "
                "round, and the chip clocks ~1.9 GHz with every CU issuing matrix instructions")
MODE_NOTE = {
    "f32": "fp32 MFMA (v_mfma_f32_16x16x4_f32), 157.3 TFLOP/s dense",
    "bf16": "bf16 MFMA on operands rounded to bf16, fp32 accumulate, 2500 TFLOP/s dense",
    "bf16x3": "bf16 MFMA on an exact 3-term bf16 split of the fp32 operands, 6 MFMAs per product block: peak = 2500 / 6 "
              "TFLOP/s of fp32-accurate products (executed MFMA rate = 6 x achieved)",
}


def make_opt(nacf_amd, method, L, V, **kw):
    base = dict(with_category=True, max_len=L, vocab_size=V, n_frames=60, fused_loss=True, beta=[0.35, 0.9], iterations=5,
                length_beam_size=6, beam_alpha=1.35, paradigm="mp")
    if method == "NACF":
        base["use_ct"] = True
    base.update(kw)
    # tuning: NACF_BENCH_OPT="key=value,key=value" overrides option keys (A/B runs inside one gpurun call: step times differ
    # by ~4 % between boxes, 2.75 vs 2.86 ms for the same build)
    for item in filter(None, os.environ.get("NACF_BENCH_OPT", "").split(",")):
        k, v = item.split("=")
        base[k] = {"true": True, "false": False}.get(v.lower(), int(v) if v.lstrip("-").isdigit() else v)
    return nacf_amd.opts.make_opt(method, "MSRVTT", **base)


def build_model(nacf_amd, opt, dev, seed=0):
    from nacf_amd import synthetic as O
    model = nacf_amd.get_model(opt)
    model.load_state_dict({k: v.clone() for k, v in O.init_state_dict(opt, seed=seed).items()})
    return model.to(dev)


def to_batch(b, dev, vw):
    out = {"feats": [f.to(dev) for f in b["feats"]], "tokens": b["tokens"].to(dev), "labels": b["labels"].to(dev),
           "category": b["category"].to(dev), "length_target": b["tgt_length"].to(dev)}
    if vw:
        out["tokens_1"], out["labels_1"] = b["tokens_1"].to(dev), b["labels_1"].to(dev)
    return out


def make_engine(model, dev, batch, ddp=None, graph="auto", eager_steps=2):
    """the step engine misc/run.py:run_train drives: launch-by-launch warm-up steps, then hipGraph replay"""
    from nacf_amd.misc.crit import get_criterion
    from nacf_amd.misc.optim import get_optimizer
    from nacf_amd.misc.run import get_forword_results
    from nacf_amd.runtime.engine import TrainStep
    crit, optim = get_criterion(model.opt), get_optimizer(model.opt, model)
    engine = TrainStep(model, crit, optim, lambda bb: get_forword_results(model.opt, model, bb, dev), ddp=ddp, graph=graph,
                       eager_steps=eager_steps)
    engine(batch)
    for _ in range(eager_steps - 1):
        engine()
    torch.cuda.synchronize()
    for _ in range(3):                            # capture + first replays
        engine()
    torch.cuda.synchronize()
    return engine, crit, optim


def timed_steps(engine, n, barrier=None):
    (barrier or torch.cuda.synchronize)()
    t0 = time.perf_counter()
    for _ in range(n):
        engine()
    (barrier or torch.cuda.synchronize)()
    return time.perf_counter() - t0


def median_step_ms(engine, min_seconds=1.0, max_steps=400):
    """per-step wall times from HIP events around single replays, repeated until >= min_seconds of timed work"""
    evs = []
    t0 = time.perf_counter()
    while len(evs) < max_steps and (len(evs) < 50 or time.perf_counter() - t0 < min_seconds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        engine()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    return {"median_ms": round(statistics.median(ms), 4), "p10_ms": round(ms[len(ms) // 10], 4),
            "p90_ms": round(ms[len(ms) * 9 // 10], 4), "steps": len(ms)}


def gemm_profile(run_once, n_prof=3):
    """HIP events around every GEMM launch of n_prof launch-by-launch passes -> per-kernel-class table"""
    from nacf_amd.runtime import ops
    ops.PROFILER.records = []
    ops.PROFILER.enabled = True
    for _ in range(n_prof):
        run_once()
    torch.cuda.synchronize()
    ops.PROFILER.enabled = False
    summ = ops.PROFILER.summary()
    ops.PROFILER.records = []
    return summ


def group_profile(run_once, n_prof=3):
    """the grouped launches of the REAL step (weight-gradient group, encoder-stream group): groups stay on, every grouped
    launch is bracketed by HIP events on the launch stream (ops.GemmProfiler.group_span)"""
    from nacf_amd.runtime import ops
    ops.PROFILER.group_records = []
    ops.PROFILER.group_enabled = True
    try:
        for _ in range(n_prof):
            run_once()
        torch.cuda.synchronize()
    finally:
        ops.PROFILER.group_enabled = False
    summ = ops.PROFILER.group_summary()
    ops.PROFILER.group_records = []
    return summ


def newest_traffic_table(leg="train"):
    """The committed PMC table of this leg from the highest round: profiles/rNN_pmc_traffic.json (training step) or
    profiles/rNN_pmc_traffic_decode.json (NA decode).  PMC counters cannot be collected inside this process; the tables come
    from separate rocprofv3 --pmc passes of this same command (tools/collect_profiles.sh, tools/pmc_traffic.py).  A leg without
    its own table reports traffic null: a kernel's bytes depend on the launch's shape, never borrowed from the other leg."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic%s.json" % ("" if leg == "train" else "_" + leg))))
    if not files:
        return None
    try:
        return files[-1], json.load(open(files[-1]))
    except (OSError, ValueError):
        return None


def roofline_from(summ, n_prof, mode, prefer_single=True, groups=None, dec_rows=None, traffic_leg="train"):
    """dominant kernel = the GEMM kernel class with the largest time per pass whose spans are ONE kernel each (spans of
    dW entry points also contain the split-K combine: listed in the table, not chosen).  `groups`: the grouped launches
    of the real step (group_profile); the classes they absorb are launched one at a time in `summ`, so the by-time
    dominant kernel of the REAL step is the larger of the two views' maxima -- in the NACF step the grouped dW kernel."""
    cands = [(k, v) for k, v in summ.items() if v["single"]] if prefer_single else list(summ.items())
    if not cands:
        cands = list(summ.items())
    if groups:
        cands = cands + list(groups.items())
    name, r = max(cands, key=lambda kv: kv[1]["ms"])
    family = "f32" if name.startswith("gemm_f32") else "bf16"
    if name.startswith("gemm_wide") or name.startswith("gemm_dma128"):
        ns = 3                       # the wide and the DMA-fed two-per-CU kernels exist in the exact mode only
    else:
        ns = 3 if (family == "bf16" and (", 3, " in name or "<3>" in name)) else 1
    kmode = "f32" if family == "f32" else ("bf16x3" if ns == 3 else "bf16")
    peak = MODE_PEAK[kmode]
    achieved = r["flops"] / (r["ms"] * 1e-3) / 1e12
    gemm_ms = sum(v["ms"] for v in summ.values()) / n_prof
    all_tf = sum(v["flops"] for v in summ.values()) / n_prof / (gemm_ms * 1e-3) / 1e12
    rl = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
          "frac": round(achieved / peak, 4), "traffic": None, "peak_basis": MODE_NOTE[kmode],
          # the same achieved rate against the two hardware peaks, so that the mode roof (2500/6) is explicit: useful flops vs
          # the fp32 matrix instruction's peak, and EXECUTED bf16 matrix flops (6 per useful one in the exact mode) vs 2.5 PF
          "frac_vs_fp32_mfma_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
          "executed_bf16_frac": round(achieved * (6.0 if kmode == "bf16x3" else 1.0) / PEAK_BF16_MFMA_TFLOPS, 4) if kmode != "f32" else None,
          "is_grouped_launch": bool(groups and name in groups),
          "launches_per_pass": r["calls"] // n_prof, "avg_launch_ms": round(r["ms"] / r["calls"], 4),
          "flops_per_pass": r["flops"] / n_prof, "all_gemm_ms_per_pass": round(gemm_ms, 3),
          "all_gemm_tflops": round(all_tf, 2), "all_gemm_frac_of_mode_peak": round(all_tf / MODE_PEAK[mode], 4),
          # rocprofv3 --pmc on this kernel family (profiles/r02_gemm_pmc_counters.txt) and the per-phase cycle stamps of
          # tools/bf16_trace.py (profiles/r02_bf16_phase_trace.txt): the matrix pipe is NOT what limits K = 512 launches
          # (the text is ONE top-level entry of the line, `limiters`: repeated per leg it pushed the decode / bf16 legs out of
          #  the tail the driver keeps)
          "limiter": ("limiters.wide" if name.startswith("gemm_wide") else "limiters.g256w" if name.startswith("g256_dw")
                      else "limiters.dma64" if name.startswith("gemm_dma64") else "limiters.dma128" if name.startswith("gemm_dma128")
                      else "limiters.tile128")}
    tab = newest_traffic_table(traffic_leg)
    if tab is not None:
        path, data = tab
        key = name[:-1] if name.endswith(">") else name      # (rocprofv3 prints trailing default template arguments: "<2, EpiArgmax, 0>")
        hits = [e for e in data.get("kernels", []) if key in e["kernel"]]
        if hits:
            n_l = sum(e["launches_sampled"] for e in hits)
            rl["traffic"] = int(sum(e["hbm_bytes"] * e["launches_sampled"] for e in hits) / n_l)
            rl["traffic_source"] = "%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes/launch, %s leg%s)" % (
                os.path.relpath(path, ROOT), traffic_leg, "; collected " + data["collected"] if data.get("collected") else "")
    # the GEMMs of the decoder layers alone (attention projections + FFN: what BASELINE's north_star quotes its matrix-core
    # utilisation target on): every launch over the decoder's rows (sequences x positions x passes) that is not the vocabulary
    # projection -- forward + dX + dW, one launch at a time
    if dec_rows:
        layer = {"flops": 0.0, "ms": 0.0, "calls": 0}
        for v in summ.values():
            for (kind, M, N, K), q in v["by_shape"].items():
                if M == dec_rows and max(N, K) <= 4096:
                    layer["flops"] += q["flops"]; layer["ms"] += q["ms"]; layer["calls"] += q["calls"]
        if layer["ms"] > 0:
            ltf = layer["flops"] / (layer["ms"] * 1e-3) / 1e12
            rl["decoder_layer_gemms"] = {"what": "attention projections + FFN GEMMs of the decoder layers (fwd + dX + dW over the %d decoder "
                                                 "rows), launched one at a time" % dec_rows,
                                         "calls_per_pass": layer["calls"] // n_prof, "ms_per_pass": round(layer["ms"] / n_prof, 3),
                                         "tflops": round(ltf, 2), "frac_of_mode_peak": round(ltf / MODE_PEAK[mode], 4)}
    table = {k: {"calls_per_pass": v["calls"] // n_prof, "ms_per_pass": round(v["ms"] / n_prof, 3),
                 "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2), "span_is_one_kernel": v["single"]}
             for k, v in summ.items()}
    return rl, table


def bench_decode(model, dev, feats, category, n_batches, with_roofline=True, mode="bf16x3"):
    from nacf_amd.models.Translator import Translator
    from nacf_amd import synthetic as O
    # decode the SEEDED INITIAL weights, not whatever the training legs left behind: at random init the predicted
    # lengths / re-masked slots -- and with them the executed work per caption -- move with the last bits of the
    # weights, so a decode of the trained-for-N-steps model is a different workload after every kernel change
    model.load_state_dict(O.init_state_dict(model.opt, seed=0))
    model.eval()
    B = feats[0].shape[0]
    tr = Translator(model, dict(model.opt), device=dev)

    def dec_once():
        with torch.no_grad():
            hyp, _ = tr.translate_batch(model.encode(feats=feats), category, None, None)
        return hyp
    for _ in range(4):       # launch by launch, hipGraph capture (decoding/na_generate.py), first replays
        hyp = dec_once()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_batches + 1)]
    t1 = time.perf_counter()
    evs[0].record()
    for i in range(n_batches):
        hyp = dec_once()
        evs[i + 1].record()
    torch.cuda.synchronize()
    ddt = (time.perf_counter() - t1) / n_batches
    per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(n_batches))
    out = {"captions_per_s": round(B / ddt, 1), "ms_per_batch": round(ddt * 1e3, 2),
           # per-batch HIP-event times: a one-off host stall (allocator growth, a graph re-instantiation) moves the mean of a
           # short loop by 10-50 %; the median says what a batch costs
           "median_ms_per_batch": round(per[len(per) // 2], 3), "max_ms_per_batch": round(per[-1], 3),
           "batch": B, "weights": "seeded init (seed 0)",
           "paradigm": "mp+ct" if model.opt.get("use_ct") else "mp", "iterations": 5, "length_beam_size": 6,
           "width": int(hyp.shape[1]), "hipgraph": any(k[0] != "seen" for k in getattr(model, "_nacf_decode_graphs", {}))}
    if with_roofline:
        # executed FLOPs / kernel times of launch-by-launch decodes (graph off) through the GEMM profiler
        tr_e = Translator(model, dict(model.opt, decode_graph="off"), device=dev)

        def eager():
            with torch.no_grad():
                tr_e.translate_batch(model.encode(feats=feats), category, None, None)
        eager()
        summ = gemm_profile(eager, n_prof=2)
        rl, table = roofline_from(summ, 2, mode, prefer_single=False, traffic_leg="decode")
        tot_f = sum(v["flops"] for v in summ.values()) / 2
        rl["executed_gflop_per_caption"] = round(tot_f / B / 1e9, 3)
        rl["as_written_gflop_per_caption"] = 17.51 if model.opt.get("use_ct") else 14.66
        rl["whole_decode_executed_tflops"] = round(tot_f / ddt / 1e12, 2)
        out["roofline"] = rl
        out["gemm_kernels"] = table
    model.train()
    return out


def bench_nab_bf16(nacf_amd, dev, B, L, V, F_):
    """BASELINE.json configs[1]: NAB, bf16 compute / fp32 master weights, batch 64, seq_len 20: train step + decode
    throughput in the bf16 mode, and against its fp32-accurate twin (same weights, same batch, mode bf16x3): logits
    error and greedy NA-token agreement (reported, not asserted -- bf16 flips argmaxes)."""
    from nacf_amd import synthetic as O
    from nacf_amd.runtime import ops
    from nacf_amd.models.Translator import Translator
    opt = make_opt(nacf_amd, "NAB", L, V)
    b = O.synth_batch(opt, B, F_, seed=3)
    res = {}
    twins = {}
    for mode in ("bf16x3", "bf16"):
        ops.set_gemm_mode(mode)
        model = build_model(nacf_amd, opt, dev)
        model.eval()
        feats, cat = [f.to(dev) for f in b["feats"]], b["category"].to(dev)
        with torch.no_grad():
            enc = model.encode(feats=feats)
            hid, *_ = model.decoder(b["tokens"].to(dev), enc_output=enc["enc_output"], category=cat)
            hid = hid[-1] if isinstance(hid, list) else hid
            logp = model.vocab_logprobs(hid)
            hyp, _ = Translator(model, dict(model.opt, decode_graph="off"), device=dev).translate_batch(enc, cat, None, None)
        twins[mode] = (logp.float().cpu(), hyp.cpu(), b["tokens"].ne(0))
        if mode == "bf16":
            model.train()
            engine, crit, optim = make_engine(model, dev, to_batch(b, dev, False), graph="on")
            dt = timed_steps(engine, 40) / 40
            summ = gemm_profile(lambda: (optim.zero_grad(), crit.get_loss(engine.forward(engine.static)).backward(),
                                         optim._optimizer.step(grad_scale=1.0)))
            rl, _ = roofline_from(summ, 3, "bf16", dec_rows=B * L)
            dec = bench_decode(model, dev, feats, cat, 10, with_roofline=False)
            res.update({"batch": B, "dtype": "bf16", "train_videos_per_s": round(B / dt, 1), "train_ms_per_step": round(dt * 1e3, 3),
                        "decode_captions_per_s": dec["captions_per_s"], "decode_ms_per_batch": dec["ms_per_batch"], "roofline": rl})
            del engine
        del model
    (lp3, hyp3, live), (lp1, hyp1, _) = twins["bf16x3"], twins["bf16"]
    w = min(hyp3.shape[1], hyp1.shape[1])
    d = (lp3 - lp1).abs()[live]
    res["vs_fp32_twin"] = {"logprob_max_abs_err": round(float(d.max()), 5), "logprob_mean_abs_err": round(float(d.mean()), 6),
                           "share_of_logprobs_within_1e-3": round(float((d <= 1e-3).float().mean()), 4),
                           "teacher_forced_argmax_agreement": round(float((lp3.argmax(-1) == lp1.argmax(-1))[live].float().mean()), 4),
                           "na_decode_token_agreement": round(float((hyp3[:, :w] == hyp1[:, :w]).float().mean()), 4),
                           "note": "random-init weights: logit margins are tiny (median 0.12, SURVEY.md section 7), so "
                                   "free-running mask-predict decodes diverge after the first flipped argmax"}
    return res


def bench_loader(args, nacf_amd, opt, dev, B, L, V, F_, engine):
    """Train-step throughput when every batch comes from nacf_amd.data.ShardLoader (synthetic shards written to a
    temp dir): per step the engine copies the loader's tensors into the graph's static input buffers and replays the
    captured step (runtime/engine.py -- the same object misc/run.py:run_train drives).  Reports the three placements of
    the shards: HBM-resident (no PCIe per step), pinned host memory (one DMA per clip, PCIe-inclusive) and memory-mapped
    file (host-thread gather into pinned staging + upload)."""
    import shutil
    import tempfile
    import numpy as np
    from nacf_amd.data import CaptionTable, FeatureShard, ShardLoader, write_feature_shard
    from nacf_amd.runtime.engine import _signature
    tmp = tempfile.mkdtemp(prefix="nacf_shards_")
    try:
        N = max(args.loader_videos, B)
        rs = np.random.RandomState(0)
        shards = []
        for m in "mi":
            path = os.path.join(tmp, "feats_%s.nacf" % m)
            write_feature_shard(path, rs.standard_normal((N, F_, 2048)).astype(np.float32))
            shards.append(FeatureShard(path))
        caps, tags, li = {}, {}, {}
        for v in range(N):
            n = int(rs.randint(4, L))
            caps["video%d" % v] = [[2] + rs.randint(6, V, size=n).tolist() + [3]]
            tags["video%d" % v] = [[2] + rs.randint(6, 12, size=n).tolist() + [3]]
            h = [0] * (L + 2)
            h[n] = 1
            li["video%d" % v] = h
        names = ["<pad>", "<unk>", "<bos>", "<eos>", "<mask>", "<vis>", "NOUN", "VERB", "DET", "ADJ", "ADP", "PRON"]
        info = dict(itow={i: "w%d" % i for i in range(V)}, itop=dict(enumerate(names)), itoc={v: v % 20 for v in range(N)},
                    length_info=li)
        lopt = dict(opt, n_frames=F_, load_feats_type=1)
        table, vids = CaptionTable.from_corpus(caps, tags, info, list(range(N)), lopt, "train")
        out = {"videos_in_shards": N, "shard_bytes": sum(s.nbytes for s in shards), "batch": B}
        for mode, placement in (("resident_hbm", "hbm"), ("pinned_host", "host"), ("mmap", "mmap")):
            ld = ShardLoader(shards, table, vids, lopt, batch_size=B, device=dev, mode="train", seed=1, placement=placement,
                             drop_last=True)

            def run(n_epochs):
                steps = 0
                for _ in range(n_epochs):
                    for b in ld:
                        b["category"] = b["category"].view(-1, 1)
                        assert _signature(b) == engine.sig, "loader batch does not match the captured step"
                        engine(b)
                        ld.bind_outputs(engine.static)     # as misc/run.py:run_train does: build the next batch in place
                        steps += 1
                return steps
            run(1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            steps = run(max(1, 24 // max(1, len(ld))))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out[mode] = {"videos_per_s": round(B * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps}
            del ld
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def bench_cpu(opt, sd, O, B, F_, budget_s=18.0):
    """The oracle (plain eager PyTorch fp32 restatement of the reference) on this box's host cores: the NACF train step
    on the SAME batch shape as the GPU leg with all cores, then with one thread on a smaller batch, and the NA decode.
    Bounded samples (the default run must finish in minutes); 1 warm-up each, median of the timed steps.  SURVEY.md 8(d)."""
    from oracle import nacf_oracle as ORACLE   # the CPU checker: imported for THIS leg only, never measured as product

    def train_leg(cb, threads, budget, min_steps):
        torch.set_num_threads(threads)
        cbatch = O.synth_batch(opt, cb, F_, seed=1)
        sd_c = {k: v.clone() for k, v in sd.items()}
        st = {}

        def step():
            return ORACLE.train_step(sd_c, dict(opt), cbatch["feats"], [cbatch["tokens_1"], cbatch["tokens"]], cbatch["category"],
                                     [cbatch["labels_1"], cbatch["labels"]], cbatch["tgt_length"], st,
                                     lr=opt["learning_rate"], training=True)
        for _ in range(N_WARM):                          # >= 3 warm-ups, >= 10 timed
            step()
        ts = []
        for _ in range(N_TIMED):
            t0 = time.perf_counter()
            step()
            ts.append(time.perf_counter() - t0)
        return cb / statistics.median(ts), len(ts), [round(cb / max(ts), 2), round(cb / min(ts), 2)]
    N_WARM, N_TIMED = 3, 10
    ncpu = os.cpu_count()
    all_threads = torch.get_num_threads()
    v_all, n_all, spread_all = train_leg(B, all_threads, budget_s, 2)
    # one thread, the SAME 128-video batch: a step takes ~5 s there, so 1 warm-up + 3 timed
    N_WARM, N_TIMED = 1, 3
    v_one, n_one, spread_one = train_leg(B, 1, budget_s * 0.6, 2)
    N_WARM, N_TIMED = 3, 10
    torch.set_num_threads(all_threads)
    # NA decode (mask-predict + coarse templates, T = 5, lbs = 6), batch 32
    db = O.synth_batch(opt, 32, F_, seed=2)
    dec = dict(paradigm="mp", use_ct=True, iterations=5, length_beam_size=6, beam_alpha=1.35)

    def dstep():
        with torch.no_grad():
            enc = ORACLE.encode(sd, opt, db["feats"], training=False)
            return ORACLE.generate(sd, opt, dec, enc, db["category"])
    for _ in range(N_WARM):
        dstep()
    ts = []
    for _ in range(N_TIMED):
        t0 = time.perf_counter()
        dstep()
        ts.append(time.perf_counter() - t0)
    return {"value": round(v_all, 2), "unit": "videos/s", "cores": all_threads, "kind": "port",
            "sample": "NACF train step (fwd+loss+bwd+clip+Adam, dropout 0.5) of %d videos, same shapes as the GPU leg: %d warm-ups + "
                      "%d timed steps, median; oracle = plain eager PyTorch fp32" % (B, N_WARM, n_all),
            "spread_min_max": spread_all,       # slowest / fastest timed step, videos/s: the box's other tenants move this 2-3x
            "cpu_model": cpu_model_name(), "logical_cpus": ncpu,
            "one_thread": {"value": round(v_one, 3), "unit": "videos/s", "cores": 1,
                           "spread_min_max": spread_one,
                           "sample": "the same %d-video step on ONE thread: 1 warm-up + %d timed steps, median (a step takes seconds "
                                     "there; 3 + 10 of them would not fit the default run)" % (B, n_one)},
            "decode": {"value": round(32 / statistics.median(ts), 2), "unit": "captions/s", "cores": all_threads,
                       "spread_min_max": [round(32 / max(ts), 2), round(32 / min(ts), 2)],
                       "sample": "oracle encode + generate (mp + coarse templates, T=5, lbs=6) of 32 videos: %d warm-ups + %d timed, "
                                 "median" % (N_WARM, len(ts))}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="videos per GPU")
    ap.add_argument("--seq-len", type=int, default=20)
    ap.add_argument("--vocab", type=int, default=10547)
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto")
    ap.add_argument("--gemm-mode", choices=["f32", "bf16x3", "bf16"], default="bf16x3",
                    help="GEMM arithmetic of the headline leg (bf16x3 = fp32-accurate split on the bf16 MFMA: the parity mode)")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--decode-only", action="store_true", help="profiling aid: a short training warm-up, then only the NA-decode leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--decode-batches", type=int, default=20)
    ap.add_argument("--no-compare", action="store_true", help="skip the extra legs (config 1 bf16, NACF bf16, L=30, config 5)")
    ap.add_argument("--no-loader", action="store_true", help="skip the shard-loader leg (SURVEY 8f row 1)")
    ap.add_argument("--loader-videos", type=int, default=1024, help="videos in the synthetic feature shards")
    args = ap.parse_args()
    if args.decode_only:      # (what tools/collect_profiles.sh profiles as the decode-only set)
        args.no_compare = args.no_loader = args.no_cpu_baseline = True
        args.no_decode = False

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # NACF_BENCH_FORCE_DIST=1: run the N>1 code path (staged backward, bucketed RCCL all-reduce on its own stream,
    # separate Adam graphs) with a 1-rank process group -- the mechanics can be exercised on a 1-GPU box
    force_dist = os.environ.get("NACF_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import nacf_amd
    from nacf_amd.runtime import ops
    from nacf_amd.runtime.ddp import DataParallel
    from nacf_amd import synthetic as O   # seeded synthetic weights / batches (input generators only)

    B, L, V, F_ = args.batch, args.seq_len, args.vocab, 60
    mode = args.gemm_mode
    ops.set_gemm_mode(mode)
    # more than one rank: BatchNorm statistics of the GLOBAL batch (three [2, 512]-class all-reduces per step), so that
    # N-rank training is one process with the global batch (models/joint_representation.py:43-45)
    opt = make_opt(nacf_amd, "NACF", L, V, sync_bn=(world > 1 or force_dist) and os.environ.get("NACF_BENCH_SYNC_BN", "1") != "0")
    sd = O.init_state_dict(opt, seed=0)
    model = build_model(nacf_amd, opt, dev)
    model.train()
    ddp = DataParallel(model, force_collectives=force_dist)
    ddp.broadcast_parameters()
    multi = world > 1 or force_dist
    n_params = sum(p.numel() for p in model.parameters())

    # each rank owns its shard of the global batch (seeded per rank), resident in HBM
    batch = to_batch(O.synth_batch(opt, B, F_, seed=1 + rank), dev, True)
    if args.decode_only:
        # profiling aid (tools/collect_profiles.sh): the NA-decode leg and nothing else on the device, so that a rocprofv3 /
        # PMC table of this command holds decode launches only (same seeded weights as the default run's decode leg).  NOT the
        # driver's line.
        model.eval()
        decode = bench_decode(model, dev, batch["feats"], batch["category"], args.decode_batches, mode=mode)
        print(json.dumps({"metric": METRIC, "leg": "decode only (profiling aid, not the driver's line)", "decode": decode}), flush=True)
        return
    engine, crit, optim = make_engine(model, dev, batch, ddp=ddp if multi else None, graph=args.graph,
                                      eager_steps=max(args.warmup, 2))
    use_graph = engine.captured
    staged = engine.staged

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    dt = timed_steps(engine, args.steps, barrier)
    if multi:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    ms_per_step = dt / args.steps * 1e3
    videos_per_s = B * world * args.steps / dt
    final_loss = float(engine.loss)
    rank_losses = None
    if multi:                                   # every rank's loss (different shards: all must be finite and close)
        ls = [torch.zeros(1, device=dev) for _ in range(world)]
        dist.all_gather(ls, engine.loss.detach().reshape(1).float())
        rank_losses = [round(float(x), 4) for x in ls]

    out = None
    if rank == 0:
        extra_timing = median_step_ms(engine) if not multi else None

        def eager_step():           # rank-local launch-by-launch step (no collective: only rank 0 is here)
            optim.zero_grad()
            crit.get_loss(engine.forward(engine.static)).backward()
            optim._optimizer.step(grad_scale=1.0)
        summ = gemm_profile(eager_step)

        def grouped_step():         # the same step as the engine runs it: weight-gradient GEMMs grouped per backward pass
            optim.zero_grad()
            loss_ = crit.get_loss(engine.forward(engine.static))
            with ops.dw_group():
                loss_.backward()
            optim._optimizer.step(grad_scale=1.0)
        groups = group_profile(grouped_step) if mode != "f32" else None
        roofline, gemm_table = roofline_from(summ, 3, mode, groups=groups, dec_rows=2 * B * L)
        if groups:
            gemm_table.update({k + " [grouped launch of the real step]":
                               {"calls_per_pass": v["calls"] // 3, "problems_per_pass": v["problems"] // 3,
                                "ms_per_pass": round(v["ms"] / 3, 3), "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                "span_is_one_kernel": True} for k, v in groups.items()})

        feats, category = batch["feats"], batch["category"]
        decode = None
        if not args.no_decode and world == 1:
            decode = bench_decode(model, dev, feats, category, args.decode_batches, mode=mode)

        loader_leg = None
        if not args.no_loader and engine.captured and not multi:
            loader_leg = bench_loader(args, nacf_amd, model.opt, dev, B, L, V, F_, engine)

        l30 = nacf_bf16 = nab = compare = None
        if not args.no_compare and world == 1:
            # ---- the reference's MSRVTT default length (opts.py:161-169: max_len 30), same model / batch size
            opt30 = make_opt(nacf_amd, "NACF", 30, V)
            m30 = build_model(nacf_amd, opt30, dev)
            m30.train()
            e30, _, _ = make_engine(m30, dev, to_batch(O.synth_batch(opt30, B, F_, seed=1), dev, True), graph="on")
            d30 = timed_steps(e30, 40) / 40
            l30 = {"seq_len": 30, "batch": B, "videos_per_s": round(B / d30, 1), "ms_per_step": round(d30 * 1e3, 3),
                   "gemm_mode": mode, "algorithmic_gflop_per_video_as_written": 5.210}
            del e30, m30

            # ---- BASELINE.json configs[4]: ARB2 beam-5 autoregressive decode vs NACF parallel decode, batch 256
            if not args.no_decode:
                from nacf_amd.models.Translator import Translator
                CB = 256
                cb = O.synth_batch(opt, CB, F_, seed=7)
                cfeats, ccat = [f.to(dev) for f in cb["feats"]], cb["category"].to(dev)

                def timed(fn, n=5):
                    for _ in range(4):
                        fn()
                    torch.cuda.synchronize()
                    t_ = time.perf_counter()
                    for _ in range(n):
                        fn()
                    torch.cuda.synchronize()
                    return (time.perf_counter() - t_) / n
                model.eval()
                tr_na = Translator(model, dict(model.opt), device=dev)

                def na():
                    with torch.no_grad():
                        return tr_na.translate_batch(model.encode(feats=cfeats), ccat, None, None)
                t_na = timed(na)
                aopt = nacf_amd.opts.make_opt("ARB2", "MSRVTT", with_category=True, max_len=L, vocab_size=V, n_frames=60,
                                              beam_size=5, beam_alpha=1.0, topk=1)
                amodel = build_model(nacf_amd, aopt, dev)
                amodel.eval()
                tr_ar = Translator(amodel, dict(amodel.opt), device=dev)

                def ar():
                    with torch.no_grad():
                        return tr_ar.translate_batch(amodel.encode(feats=cfeats), ccat, None, None)
                t_ar = timed(ar)
                compare = {"batch": CB, "nacf_mp_ct_captions_per_s": round(CB / t_na, 1),
                           "arb2_beam5_captions_per_s": round(CB / t_ar, 1), "nacf_over_arb2": round(t_ar / t_na, 2),
                           "gemm_mode": mode,
                           "note": "random-init weights: AR hypotheses rarely emit <eos>, so beam search runs all max_len-1 steps (each on the last "
                                   "slot of every hypothesis only, DESIGN.md 4f); the NA side decodes 6 length candidates in 6 passes"}
                del amodel
                model.train()

            # ---- bf16 throughput mode: NACF (same workload as the headline) and BASELINE.json configs[1] (NAB, batch 64)
            ops.set_gemm_mode("bf16")
            mb = build_model(nacf_amd, opt, dev)
            mb.train()
            eb, cb_, ob_ = make_engine(mb, dev, batch, graph="on")
            db_ = timed_steps(eb, 40) / 40
            sb = gemm_profile(lambda: (ob_.zero_grad(), cb_.get_loss(eb.forward(eb.static)).backward(),
                                       ob_._optimizer.step(grad_scale=1.0)))
            rlb, _ = roofline_from(sb, 3, "bf16", dec_rows=2 * B * L)
            decb = bench_decode(mb, dev, feats, category, 10, with_roofline=False) if not args.no_decode else None
            nacf_bf16 = {"dtype": "bf16", "batch": B, "videos_per_s": round(B / db_, 1), "ms_per_step": round(db_ * 1e3, 3),
                         "final_loss": round(float(eb.loss), 4), "roofline": rlb,
                         "decode_captions_per_s": decb["captions_per_s"] if decb else None,
                         "note": "fp32 master weights / activations in HBM, every GEMM operand rounded to bf16 (RNE), fp32 accumulate"}
            del eb, mb
            nab = bench_nab_bf16(nacf_amd, dev, 64, L, V, F_)
            ops.set_gemm_mode(mode)

        cpu = None
        if not args.no_cpu_baseline and world == 1:      # reported at N=1 only (the other ranks would sit in a barrier)
            cpu = bench_cpu(opt, sd, O, B, F_)

        out = {"metric": METRIC, "value": round(videos_per_s, 1), "unit": "videos/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if mode != "bf16" else "bf16",
               "data": "synthetic",
               "config": {"workload": "NACF train step, MSRVTT-shape (configs[2]/[3]): %d videos/GPU, 2x60x2048 fp32 "
                                      "feats, seq_len %d, V=%d, dropout 0.5, Adam" % (B, L, V),
                          "gemm_mode": mode, "arithmetic": MODE_NOTE[mode],
                          "global_batch": B * world, "seq_len": L, "vocab": V, "params": n_params,
                          "parallelism": "dp%d" % world, "hipgraph": bool(use_graph), "live_row_gemms": True,
                          "sync_bn": bool(model.opt.get("sync_bn", False)),
                          "overlapped_allreduce": bool(staged),
                          "graph_collectives": bool(getattr(engine, "graph_collectives", False)),
                          "gradient_buckets": (3 if engine.three else 2) if staged else 1},
               "timing": extra_timing, "rank_losses": rank_losses,
               # (bulky / referenced entries first: what reads only the END of this line keeps the legs below)
               "gemm_kernels": gemm_table, "limiters": {"tile128": LIMITER_128, "wide": LIMITER_WIDE, "g256w": LIMITER_G256W, "dma64": LIMITER_DMA64, "dma128": LIMITER_DMA128},
               "loader_fed": loader_leg, "train_L30": l30, "config5_ar_vs_na": compare,
               "nacf_bf16": nacf_bf16, "config1_nab_bf16": nab, "decode": decode,
               "roofline": roofline, "cpu_baseline": cpu, "final_loss": round(final_loss, 4)}
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes a version banner to the C-level stdout; flush it first so the JSON line is the LAST line
        import ctypes
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        # The full record (every leg with its tables and prose) goes to stderr and to gpurun_out/bench_full.json; the ONE line on
        # stdout is the contract line plus the headline number of every leg, short enough to survive a 2000-character tail
        # (the decode half of the metric did not reach the driver's record)
        full = json.dumps(out)
        print(full, file=sys.stderr, flush=True)
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_full.json"), "w") as f:
                f.write(full + "\n")
        except OSError:
            pass
        print(json.dumps(compact_line(out)), flush=True)


def compact_line(out):
    """the driver's line: contract keys, `roofline`, `cpu_baseline`, and per leg only the numbers"""
    def pick(d, keys):
        return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None} if isinstance(d, dict) else None

    def roof(r):
        return pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "all_gemm_tflops",
                        "all_gemm_frac_of_mode_peak"))
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data") if k in out}
    cfg = out.get("config") or {}
    line["config"] = {"workload": "NACF train step, MSRVTT-shape: %s videos/GPU, 2x60x2048 fp32, L=%s, V=%s, dropout 0.5, Adam"
                                  % (cfg.get("global_batch", 0) // max(1, out.get("n_gpus", 1)), cfg.get("seq_len"), cfg.get("vocab")),
                      "gemm_mode": cfg.get("gemm_mode"), "global_batch": cfg.get("global_batch"), "seq_len": cfg.get("seq_len"),
                      "parallelism": cfg.get("parallelism"), "hipgraph": cfg.get("hipgraph"), "sync_bn": cfg.get("sync_bn"),
                      "overlapped_allreduce": cfg.get("overlapped_allreduce"), "graph_collectives": cfg.get("graph_collectives"),
                      "gradient_buckets": cfg.get("gradient_buckets")}
    if out.get("rank_losses") is not None:
        line["rank_losses"] = out["rank_losses"]
    line["final_loss"] = out.get("final_loss")
    if isinstance(out.get("timing"), dict):
        line["timing"] = pick(out["timing"], ("median_ms",))
    dec = out.get("decode")
    if dec:
        line["decode"] = pick(dec, ("captions_per_s", "ms_per_batch", "batch", "paradigm", "iterations", "length_beam_size"))
        line["decode"]["roofline"] = pick(dec.get("roofline"), ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"))
    nb = out.get("nacf_bf16")
    if nb:
        line["nacf_bf16"] = pick(nb, ("videos_per_s", "ms_per_step", "decode_captions_per_s"))
        line["nacf_bf16"]["frac"] = (nb.get("roofline") or {}).get("frac")
    nab = out.get("config1_nab_bf16")
    if nab:
        line["config1_nab_bf16"] = pick(nab, ("batch", "train_videos_per_s", "train_ms_per_step", "decode_captions_per_s"))
        line["config1_nab_bf16"]["frac"] = (nab.get("roofline") or {}).get("frac")
    c5 = out.get("config5_ar_vs_na")
    if c5:
        line["config5_ar_vs_na"] = pick(c5, ("batch", "nacf_mp_ct_captions_per_s", "arb2_beam5_captions_per_s", "nacf_over_arb2"))
    lf = out.get("loader_fed")
    if lf:      # (videos/s per placement; train_L30 and every table stay in the full record)
        line["loader_fed"] = {k: (v or {}).get("videos_per_s") for k, v in lf.items() if isinstance(v, dict)}
    line["roofline"] = roof(out.get("roofline"))
    cpu = out.get("cpu_baseline")
    if cpu:
        line["cpu_baseline"] = pick(cpu, ("value", "unit", "cores", "kind"))
        line["cpu_baseline"]["sample"] = "oracle train step, 128 videos, median of 10"
        if isinstance(cpu.get("one_thread"), dict):
            line["cpu_baseline"]["one_thread_value"] = cpu["one_thread"].get("value")
        if isinstance(cpu.get("decode"), dict):
            line["cpu_baseline"]["decode_captions_per_s"] = cpu["decode"].get("value")
    line["details"] = "full record: stderr"
    return line


if __name__ == "__main__":
    main()
