"""NA decode (mask-predict + coarse templates, B=128) in a loop, for rocprofv3 --stats (tuning aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nacf_amd
from nacf_amd.models.Translator import Translator
from nacf_amd import synthetic as S
dev = torch.device("cuda:0")
opt = nacf_amd.opts.make_opt("NACF", "MSRVTT", with_category=True, max_len=20, vocab_size=10547, fused_loss=True)
m = nacf_amd.get_model(opt); m.load_state_dict(S.init_state_dict(opt, 0)); m.to(dev).eval()
B = int(os.environ.get('DECODE_BATCH', '128'))
b = S.synth_batch(opt, B, 60, seed=1)
feats = [f.to(dev) for f in b["feats"]]; cat = b["category"].to(dev)
dopt = dict(m.opt); dopt.update(paradigm=os.environ.get("DECODE_PARADIGM", "mp"), use_ct=True, iterations=5, length_beam_size=6, beam_alpha=1.35,
                                q=int(os.environ.get("DECODE_Q", "1")))
tr = Translator(m, dopt, device=dev)
def once():
    with torch.no_grad():
        enc = m.encode(feats=feats)
        return tr.translate_batch(enc, cat, None, None)
for _ in range(2): once()
torch.cuda.synchronize(); t = time.perf_counter()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for _ in range(n): once()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / n
print("ms per batch", dt * 1e3, "captions/s", B / dt)
