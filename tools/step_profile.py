"""A captured training step in a loop, for rocprofv3 --stats:  METHOD=NAB BATCH=64 MODE=bf16 python tools/step_profile.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nacf_amd
from nacf_amd import synthetic as S
from nacf_amd.misc.crit import get_criterion
from nacf_amd.misc.optim import get_optimizer
from nacf_amd.misc.run import get_forword_results
from nacf_amd.runtime.engine import TrainStep
from nacf_amd.runtime import ops
dev = torch.device("cuda:0")
method, B, mode = os.environ.get("METHOD", "NAB"), int(os.environ.get("BATCH", "64")), os.environ.get("MODE", "bf16")
ops.set_gemm_mode(mode)
opt = nacf_amd.opts.make_opt(method, "MSRVTT", with_category=True, max_len=20, vocab_size=10547, n_frames=60, fused_loss=True)
m = nacf_amd.get_model(opt); m.load_state_dict(S.init_state_dict(opt, 0)); m.to(dev).train()
crit, optim = get_criterion(m.opt), get_optimizer(m.opt, m)
b = S.synth_batch(opt, B, 60, seed=2)
batch = {"feats": [f.to(dev) for f in b["feats"]], "tokens": b["tokens"].to(dev), "labels": b["labels"].to(dev),
         "category": b["category"].to(dev), "length_target": b["tgt_length"].to(dev)}
if method == "NACF":
    batch["tokens_1"], batch["labels_1"] = b["tokens_1"].to(dev), b["labels_1"].to(dev)
engine = TrainStep(m, crit, optim, lambda bb: get_forword_results(m.opt, m, bb, dev), graph="on")
engine(batch)
for _ in range(6):
    engine()
torch.cuda.synchronize(); t = time.perf_counter()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for _ in range(n):
    engine()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / n
print("captured", engine.captured, "ms per step", dt * 1e3, "videos/s", B / dt)
