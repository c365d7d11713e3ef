import os, sys
sys.path.insert(0, "/root/repo")
import torch
import nacf_amd
from nacf_amd.misc.crit import get_criterion
from nacf_amd.misc.optim import get_optimizer
from nacf_amd import synthetic as O
dev = torch.device("cuda:0")
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
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    step()
torch.cuda.synchronize()
import collections
agg = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::add", "aten::add_", "aten::mul", "aten::mul_", "aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::zeros_like", "aten::div", "aten::sum", "aten::cat", "aten::mean", "aten::contiguous", "aten::clone", "aten::sub", "aten::neg", "aten::masked_fill", "aten::eq", "aten::ne", "aten::to", "aten::_to_copy"):
        st = [s for s in (ev.stack or []) if "nacf" in s or "non-autoreg" in s]
        agg[(ev.name, str(ev.input_shapes)[:60], st[0][-70:] if st else "")] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(v, k)
