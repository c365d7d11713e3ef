"""Helper of tests/test_bench_contract_gpu.py (run as a subprocess: a process group is per process): N training steps of a small
NACF model with dropout 0, through the step engine, in one of the launch sequences bench.py --gpus N uses; the post-step flat
weights go to the file named on the command line.

  python tests/ddp_step_helper.py out.pt single            one graph, no collective
  python tests/ddp_step_helper.py out.pt dist              1-rank RCCL group forced: staged backward, bucketed all-reduce, SyncBN
  (NACF_DDP_STAGES=3, NACF_DDP_GRAPH_COLLECTIVES=1 select the three-bucket / in-graph variants as in bench.py)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    out_path, kind = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    multi = kind == "dist"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import nacf_amd
    from nacf_amd import synthetic as S
    from nacf_amd.misc.crit import get_criterion
    from nacf_amd.misc.optim import get_optimizer
    from nacf_amd.misc.run import get_forword_results
    from nacf_amd.runtime.ddp import DataParallel
    from nacf_amd.runtime.engine import TrainStep
    B, L, V, F_ = 16, 20, 1001, 12
    opt = nacf_amd.opts.make_opt("NACF", "MSRVTT", with_category=True, max_len=L, vocab_size=V, n_frames=F_, fused_loss=True,
                                 hidden_dropout_prob=0.0, encoder_dropout=0.0, use_ct=True, sync_bn=multi)
    model = nacf_amd.get_model(opt)
    model.load_state_dict(S.init_state_dict(opt, 0))
    model.to(dev).train()
    ddp = DataParallel(model, force_collectives=multi)
    ddp.broadcast_parameters()
    b = S.synth_batch(opt, B, F_, seed=5)
    batch = {"feats": [f.to(dev) for f in b["feats"]], "tokens": b["tokens"].to(dev), "labels": b["labels"].to(dev),
             "category": b["category"].to(dev), "length_target": b["tgt_length"].to(dev), "tokens_1": b["tokens_1"].to(dev),
             "labels_1": b["labels_1"].to(dev)}
    crit, optim = get_criterion(model.opt), get_optimizer(model.opt, model)
    # eager_steps = 1: step 1 launch by launch, steps 2.. replayed from the captured graph(s)
    engine = TrainStep(model, crit, optim, lambda bb: get_forword_results(model.opt, model, bb, dev), ddp=ddp if multi else None,
                       graph="on", eager_steps=1)
    engine(batch)
    for _ in range(steps - 1):
        engine()
    torch.cuda.synchronize()
    torch.save({"weights": model.flat.data.detach().cpu(), "loss": float(engine.loss), "captured": bool(engine.captured),
                "staged": bool(engine.staged), "buckets": (3 if engine.three else 2) if engine.staged else 1,
                "graph_collectives": bool(getattr(engine, "graph_collectives", False))}, out_path)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
