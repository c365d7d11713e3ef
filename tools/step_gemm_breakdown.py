"""Per-launch GEMM breakdown of one NACF train step (tuning aid, GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nacf_amd
from nacf_amd.misc.crit import get_criterion
from nacf_amd.misc.optim import get_optimizer
from nacf_amd.runtime import ops
from nacf_amd import synthetic as O
dev = torch.device("cuda:0")
if len(sys.argv) > 1:
    ops.set_gemm_mode(sys.argv[1])
opt = nacf_amd.opts.make_opt("NACF", "MSRVTT", with_category=True, max_len=20, vocab_size=10547, fused_loss=True, beta=[0.35, 0.9])
m = nacf_amd.get_model(opt); m.load_state_dict(O.init_state_dict(opt, 0)); m.to(dev).train()
crit, optim = get_criterion(m.opt), get_optimizer(m.opt, m)
b = O.synth_batch(opt, 128, 60, seed=1)
feats = [f.to(dev) for f in b["feats"]]; toks = [b["tokens_1"].to(dev), b["tokens"].to(dev)]
labs = [b["labels_1"].to(dev), b["labels"].to(dev)]; cat = b["category"].to(dev); tl = b["tgt_length"].to(dev)
def step():
    optim.zero_grad()
    r = m(feats=feats, tgt_tokens=toks, category=cat); r["tgt_word_labels"] = labs; r["tgt_length"] = tl
    crit.get_loss(r).backward(); optim.step()
for _ in range(3): step()
torch.cuda.synchronize()
ops.PROFILER.enabled = True
step(); torch.cuda.synchronize()
ops.PROFILER.enabled = False
tot = 0
print("%-62s %-20s %6s %8s %7s" % ("kernel", "M,N,K", "live", "ms", "TF"))
KIND = {0: "fwd", 1: "dX", 2: "dW"}
for name, shape, a, e, single, rows, kind in ops.PROFILER.records:
    M, N, K = shape
    live = min(M, int(rows.count)) if rows is not None else M
    ms = a.elapsed_time(e); tot += ms
    print("%-3s %-50s %-20s %6d %8.3f %7.1f" % (KIND[kind], name[name.index("<"):], "%d,%d,%d" % shape, live, ms, 2.0 * live * N * K / ms / 1e9))
print("total gemm ms", tot)
