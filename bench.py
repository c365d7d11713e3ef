#!/usr/bin/env python
"""bench.py -- the hot path of BASELINE.json on N GPUs of one node.

  python bench.py --gpus N --steps K --warmup W          (N=1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2]/[3]): NACF, MSRVTT-shape
synthetic batch -- 128 videos per GPU, motion+image features 60x2048 fp32 each
(U[0,1)), seq_len 20, V = 10547, category embeddings, dropout 0.5 as in the
reference defaults, random-init weights.  One "step" = zero_grad + forward (two
decoder passes) + fused loss + backward + [RCCL all-reduce of the flat gradient
bucket] + clip(+-5) + Adam, inputs resident in HBM.  value = global videos/s.
Also reported (same JSON line): NA-decode captions/s (mask-predict + coarse
templates, T=5, lbs=6), the live roofline of the dominant GEMM kernel (HIP
events around every launch of it) and the CPU baseline (the oracle timed on the
host cores, bounded sample).  Arithmetic is fp32 on the MFMA f32 path (exact
parity mode).  Nothing is cached across steps; the only work not executed is
work whose result is identically zero in the reference too: decoder rows whose
token is <pad> and vocabulary rows without a label (live-row GEMMs, DESIGN.md
section 4) -- gradients, loss and decoded tokens are unchanged (parity tests run this
same path).  The roofline counts only the FLOPs actually executed.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

METRIC = "training videos/sec (whole node) + NA-decode captions/sec, NACF MSRVTT-shape"
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_*_f32 dense peak


def make_opt(nacf_amd, L, V):
    return nacf_amd.opts.make_opt("NACF", "MSRVTT", with_category=True, max_len=L, vocab_size=V, n_frames=60,
                                  fused_loss=True, beta=[0.35, 0.9], use_ct=True, iterations=5, length_beam_size=6,
                                  beam_alpha=1.35, paradigm="mp")


def bench_nab(nacf_amd, dev, B, L, V, F_):
    """train-step and decode throughput of NAB (BASELINE.json configs[1]) with the step captured in a hipGraph"""
    from nacf_amd.misc.crit import get_criterion
    from nacf_amd.misc.optim import get_optimizer
    from nacf_amd.models.Translator import Translator
    from nacf_amd import synthetic as O
    opt = nacf_amd.opts.make_opt("NAB", "MSRVTT", with_category=True, max_len=L, vocab_size=V, n_frames=60, fused_loss=True,
                                 beta=[0.35, 0.9], iterations=5, length_beam_size=6, beam_alpha=1.35, paradigm="mp")
    model = nacf_amd.get_model(opt)
    model.load_state_dict({k: v.clone() for k, v in O.init_state_dict(opt, seed=0).items()})
    model.to(dev).train()
    crit, optim = get_criterion(model.opt), get_optimizer(model.opt, model)
    b = O.synth_batch(opt, B, F_, seed=3)
    feats = [f.to(dev) for f in b["feats"]]
    tokens, labels = b["tokens"].to(dev), b["labels"].to(dev)
    category, tgt_length = b["category"].to(dev), b["tgt_length"].to(dev)

    from nacf_amd.misc.run import get_forword_results
    from nacf_amd.runtime.engine import TrainStep
    engine = TrainStep(model, crit, optim, lambda bb: get_forword_results(model.opt, model, bb, dev), graph="on")
    engine({"feats": feats, "tokens": tokens, "labels": labels, "category": category, "length_target": tgt_length})
    for _ in range(5):          # two more launch-by-launch steps, capture, replays
        engine()
    assert engine.captured
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        engine()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    model.eval()
    tr = Translator(model, dict(model.opt), device=dev)

    def dec():
        with torch.no_grad():
            return tr.translate_batch(model.encode(feats=feats), category, None, None)
    for _ in range(4):
        dec()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(10):
        dec()
    torch.cuda.synchronize()
    ddt = (time.perf_counter() - t1) / 10
    return {"batch": B, "train_videos_per_s": round(B / dt, 1), "train_ms_per_step": round(dt * 1e3, 3),
            "decode_captions_per_s": round(B / ddt, 1), "decode_ms_per_batch": round(ddt * 1e3, 2), "dtype": "f32"}


def bench_loader(args, nacf_amd, opt, dev, B, L, V, F_, engine):
    """Train-step throughput when every batch comes from nacf_amd.data.ShardLoader (synthetic shards written to a
    temp dir): per step the engine copies the loader's tensors into the graph's static input buffers and replays the
    captured step (runtime/engine.py -- the same object misc/run.py:run_train drives).  Reports the three placements of the shards: HBM-resident (no PCIe per step), pinned host memory (one
    DMA per clip, PCIe-inclusive) and memory-mapped file (host-thread gather into pinned staging + upload)."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="videos per GPU")
    ap.add_argument("--seq-len", type=int, default=20)
    ap.add_argument("--vocab", type=int, default=10547)
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--decode-batches", type=int, default=20)
    ap.add_argument("--no-compare", action="store_true", help="skip the ARB2 beam-5 vs NACF decode comparison (config 5)")
    ap.add_argument("--no-loader", action="store_true", help="skip the shard-loader leg (SURVEY 8f row 1)")
    ap.add_argument("--loader-videos", type=int, default=1024, help="videos in the synthetic feature shards")
    ap.add_argument("--gemm-mode", choices=["f32", "bf16x3", "bf16"], default=None,
                    help="GEMM arithmetic of the headline leg (default: the library default)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # NACF_BENCH_FORCE_DIST=1: run the N>1 code path (staged backward, bucketed RCCL all-reduce on its own stream,
    # separate Adam graph) with a 1-rank process group -- the mechanics can be exercised on a 1-GPU box
    force_dist = os.environ.get("NACF_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import nacf_amd
    from nacf_amd.misc.crit import get_criterion
    from nacf_amd.misc.optim import get_optimizer
    from nacf_amd.models.Translator import Translator
    from nacf_amd.runtime import ops
    from nacf_amd.runtime.ddp import DataParallel
    from nacf_amd import synthetic as O   # seeded synthetic weights / batches (input generators only)

    B, L, V, F_ = args.batch, args.seq_len, args.vocab, 60
    if args.gemm_mode is not None:
        ops.set_gemm_mode(args.gemm_mode)
    opt = make_opt(nacf_amd, L, V)
    sd = O.init_state_dict(opt, seed=0)
    model = nacf_amd.get_model(opt)
    model.load_state_dict({k: v.clone() for k, v in sd.items()})
    model.to(dev).train()
    ddp = DataParallel(model, force_collectives=force_dist)
    ddp.broadcast_parameters()
    multi = world > 1 or force_dist
    crit = get_criterion(model.opt)
    optim = get_optimizer(model.opt, model)
    n_params = sum(p.numel() for p in model.parameters())

    # each rank owns its shard of the global batch (seeded per rank), resident in HBM
    batch = O.synth_batch(opt, B, F_, seed=1 + rank)
    feats = [f.to(dev) for f in batch["feats"]]
    tokens = [batch["tokens_1"].to(dev), batch["tokens"].to(dev)]
    labels = [batch["labels_1"].to(dev), batch["labels"].to(dev)]
    category = batch["category"].to(dev)
    tgt_length = batch["tgt_length"].to(dev)
    from nacf_amd.misc.run import get_forword_results
    from nacf_amd.runtime.engine import TrainStep

    # the step engine misc/run.py:run_train drives: launch-by-launch warm-up steps, then ONE hipGraph per step
    # (N > 1: backward in two graphs so the decoder-side gradient bucket (61 of 74 MB) is all-reduced while the
    # encoder's backward still runs, then the Adam graph -- runtime/engine.py, runtime/ddp.py)
    n_eager = max(args.warmup, 2)
    engine = TrainStep(model, crit, optim, lambda bb: get_forword_results(model.opt, model, bb, dev),
                       ddp=ddp if multi else None, graph=args.graph, eager_steps=n_eager)
    staged = engine.staged
    engine({"feats": feats, "tokens_1": tokens[0], "tokens": tokens[1], "labels_1": labels[0], "labels": labels[1],
            "category": category, "length_target": tgt_length})
    for _ in range(n_eager - 1):                  # untimed warm-up (also grows workspaces before capture)
        engine()
    torch.cuda.synchronize()
    for _ in range(3):                            # capture + first replays
        engine()
    use_graph = engine.captured
    step = engine
    loss_buf = engine.loss

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if multi:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    ms_per_step = dt / args.steps * 1e3
    videos_per_s = B * world * args.steps / dt
    final_loss = float(loss_buf)

    out = None
    if rank == 0:
        # ---- live roofline: HIP events around every GEMM launch of a few eager steps
        ops.PROFILER.enabled = True
        n_prof = 3
        for _ in range(n_prof):                  # rank-local launch-by-launch steps (no collective: only rank 0 is here)
            optim.zero_grad()
            crit.get_loss(get_forword_results(model.opt, model, engine.static, dev)).backward()
            optim._optimizer.step(grad_scale=1.0)
        torch.cuda.synchronize()
        ops.PROFILER.enabled = False
        summ = ops.PROFILER.summary()
        ops.PROFILER.records = []
        # dominant kernel = the single-launch GEMM class with the largest total time (spans of the dW
        # entry point also contain the split-K combine and bias column-sum kernels: listed, not chosen)
        dom = max(((k, v) for k, v in summ.items() if v["single"]), key=lambda kv: kv[1]["ms"])
        name, r = dom
        achieved = r["flops"] / (r["ms"] * 1e-3) / 1e12
        gemm_ms = sum(v["ms"] for v in summ.values()) / n_prof
        roofline = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                    "launches_per_step": r["calls"] // n_prof,
                    "avg_launch_ms": round(r["ms"] / r["calls"], 4),
                    "flops_per_step": r["flops"] / n_prof,
                    "all_gemm_ms_per_step": round(gemm_ms, 3),
                    "all_gemm_tflops": round(sum(v["flops"] for v in summ.values()) / n_prof / (gemm_ms * 1e-3) / 1e12, 2),
                    # not measured in this run: tools/gemm_trace.py (s_memtime vs wall_clock64 inside the workgroups)
                    "note": "peak is the 2.4 GHz paper figure; under this kernel's load the shader clock was measured at "
                            "2.05-2.13 GHz (134-140 TF at full MFMA issue), see DESIGN.md section 4"}
        # HBM traffic of that kernel: PMC counters cannot be collected from inside this process, so the
        # figure comes from the committed rocprofv3 --pmc passes of this same command (tools/pmc_traffic.py),
        # averaged per launch over all launches of the kernel; null if the table is missing.
        try:
            tab = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            hits = [e for e in tab["kernels"] if e["kernel"].startswith("void " + name)]
            if hits:
                n_l = sum(e["launches_sampled"] for e in hits)
                roofline["traffic"] = int(sum(e["hbm_bytes"] * e["launches_sampled"] for e in hits) / n_l)
                roofline["traffic_source"] = "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes/launch)"
        except (OSError, ValueError, KeyError):
            pass
        gemm_table = {k: {"calls_per_step": v["calls"] // n_prof, "ms_per_step": round(v["ms"] / n_prof, 3),
                          "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                          "span_is_one_kernel": v["single"]} for k, v in summ.items()}

        # ---- NA decode throughput (captions/s incl. encode), same model in eval mode
        decode = None
        if not args.no_decode and world == 1:
            model.eval()
            tr = Translator(model, dict(model.opt), device=dev)
            def dec_once():
                with torch.no_grad():
                    enc = model.encode(feats=feats)
                    hyp, _ = tr.translate_batch(enc, category, None, None)
                return hyp
            for _ in range(4):       # launch by launch, hipGraph capture (decoding/na_generate.py), first replays
                dec_once()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.decode_batches):
                hyp = dec_once()
            torch.cuda.synchronize()
            ddt = (time.perf_counter() - t1) / args.decode_batches
            decode = {"captions_per_s": round(B / ddt, 1), "ms_per_batch": round(ddt * 1e3, 2), "batch": B,
                      "paradigm": "mp+ct", "iterations": 5, "length_beam_size": 6, "width": int(hyp.shape[1]),
                      "hipgraph": any(k[0] != "seen" for k in getattr(model, "_nacf_decode_graphs", {}))}
            model.train()

        # ---- the same step fed by the shard loader (features gathered / frame-sampled / masked on
        # the device) instead of one resident synthetic batch; "streaming" includes the PCIe upload of every batch
        loader_leg = None
        if not args.no_loader and engine.captured and not multi:
            loader_leg = bench_loader(args, nacf_amd, model.opt, dev, B, L, V, F_, engine)

        # ---- BASELINE.json configs[4]: ARB2 beam-5 autoregressive decode vs NACF parallel decode, batch 256
        compare = None
        if not args.no_compare and not args.no_decode and world == 1:
            CB = 256
            cb = O.synth_batch(opt, CB, F_, seed=7)
            cfeats = [f.to(dev) for f in cb["feats"]]
            ccat = cb["category"].to(dev)
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
            amodel = nacf_amd.get_model(aopt)
            amodel.load_state_dict({k: v.clone() for k, v in O.init_state_dict(aopt, seed=0).items()})
            amodel.to(dev).eval()
            tr_ar = Translator(amodel, dict(amodel.opt), device=dev)
            def ar():
                with torch.no_grad():
                    return tr_ar.translate_batch(amodel.encode(feats=cfeats), ccat, None, None)
            t_ar = timed(ar)
            compare = {"batch": CB, "nacf_mp_ct_captions_per_s": round(CB / t_na, 1),
                       "arb2_beam5_captions_per_s": round(CB / t_ar, 1), "nacf_over_arb2": round(t_ar / t_na, 2),
                       "note": "random-init weights: AR hypotheses rarely emit <eos>, so beam search runs all max_len-1 steps"}
            del amodel
            model.train()

        # ---- BASELINE.json configs[1]: NAB (single-pass masked-LM decoder), MSRVTT-shape, batch 64, seq_len 20 --
        # the same engine on the other NA model family (fp32 here: see DESIGN.md on bf16 and greedy-token parity)
        nab = None
        if not args.no_compare and world == 1:
            nab = bench_nab(nacf_amd, dev, 64, L, V, F_)

        # ---- CPU baseline: the oracle (plain eager PyTorch fp32 restatement) on this box's host cores
        cpu = None
        if not args.no_cpu_baseline and world == 1:      # reported at N=1 only (the other ranks would sit in a barrier)
            from oracle import nacf_oracle as ORACLE   # the CPU checker: imported for THIS leg only, never measured as product
            cb = 32
            cbatch = O.synth_batch(opt, cb, F_, seed=1)
            sd_c = {k: v.clone() for k, v in sd.items()}
            st = {}
            copt = dict(opt)
            def cpu_step():
                return ORACLE.train_step(sd_c, copt, cbatch["feats"], [cbatch["tokens_1"], cbatch["tokens"]],
                                    cbatch["category"], [cbatch["labels_1"], cbatch["labels"]],
                                    cbatch["tgt_length"], st, lr=opt["learning_rate"], training=True)
            cpu_step()
            t2 = time.perf_counter()
            n_cpu = 0
            while n_cpu < 2 or (time.perf_counter() - t2 < 10 and n_cpu < 20):
                cpu_step(); n_cpu += 1
            cdt = (time.perf_counter() - t2) / n_cpu
            cpu = {"value": round(cb / cdt, 2), "unit": "videos/s", "cores": torch.get_num_threads(), "kind": "port",
                   "sample": "%d NACF train steps (fwd+loss+bwd+clip+Adam, dropout 0.5) of %d videos, same shapes; "
                             "oracle = plain eager PyTorch fp32, host has %d logical CPUs" % (n_cpu, cb, os.cpu_count())}

        out = {"metric": METRIC, "value": round(videos_per_s, 1), "unit": "videos/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "NACF train step, MSRVTT-shape (configs[2]/[3]): %d videos/GPU, 2x60x2048 fp32 "
                                      "feats, seq_len %d, V=%d, dropout 0.5, Adam" % (B, L, V),
                          "global_batch": B * world, "seq_len": L, "vocab": V, "params": n_params,
                          "parallelism": "dp%d" % world, "hipgraph": bool(use_graph), "live_row_gemms": True,
                          "overlapped_allreduce": bool(staged),
                          "gradient_buckets": (3 if engine.three else 2) if staged else 1},
               "roofline": roofline, "cpu_baseline": cpu, "decode": decode, "config5_ar_vs_na": compare,
               "loader_fed": loader_leg, "config1_nab": nab,
               "final_loss": round(final_loss, 4),
               "gemm_kernels": gemm_table}
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
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
