"""Per-launch GEMM breakdown of one NA decode batch (tuning aid, GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nacf_amd
from nacf_amd.models.Translator import Translator
from nacf_amd.runtime import ops
from nacf_amd import synthetic as S
dev = torch.device("cuda:0")
opt = nacf_amd.opts.make_opt("NACF", "MSRVTT", with_category=True, max_len=20, vocab_size=10547, fused_loss=True)
m = nacf_amd.get_model(opt); m.load_state_dict(S.init_state_dict(opt, 0)); m.to(dev).eval()
b = S.synth_batch(opt, 128, 60, seed=1)
feats = [f.to(dev) for f in b["feats"]]; cat = b["category"].to(dev)
dopt = dict(m.opt); dopt.update(paradigm="mp", use_ct=True, iterations=5, length_beam_size=6, beam_alpha=1.35, decode_graph="off")
tr = Translator(m, dopt, device=dev)
def once():
    with torch.no_grad():
        return tr.translate_batch(m.encode(feats=feats), cat, None, None)
for _ in range(2): once()
torch.cuda.synchronize()
ops.PROFILER.enabled = True
once(); torch.cuda.synchronize()
ops.PROFILER.enabled = False
import collections
agg = collections.OrderedDict()
for name, shape, a, e, single, rows, _kind in ops.PROFILER.records:
    M, N, K = shape
    live = min(M, int(rows.count)) if rows is not None else M
    k = (name[16:40], shape)
    r = agg.setdefault(k, [0, 0.0, 0.0])
    r[0] += 1; r[1] += a.elapsed_time(e); r[2] += 2.0 * live * N * K
tot = 0
for (name, shape), (n, ms, fl) in agg.items():
    tot += ms
    print("%-26s %-20s x%-3d %8.3f ms  %6.1f TF (live rows avg %d)" % (name, "%d,%d,%d" % shape, n, ms, fl / ms / 1e9, fl / n / (2.0 * shape[1] * shape[2])))
print("total gemm ms", tot)
