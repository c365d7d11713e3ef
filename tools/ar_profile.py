"""ARB2 beam-5 decode (B=256, the config-5 comparator) in a loop, for rocprofv3 --stats (tuning aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nacf_amd
from nacf_amd.models.Translator import Translator
from nacf_amd import synthetic as S
dev = torch.device("cuda:0")
B = int(os.environ.get('DECODE_BATCH', '256'))
opt = nacf_amd.opts.make_opt("ARB2", "MSRVTT", with_category=True, max_len=20, vocab_size=10547, n_frames=60, beam_size=5, beam_alpha=1.0, topk=1)
m = nacf_amd.get_model(opt); m.load_state_dict(S.init_state_dict(opt, 0)); m.to(dev).eval()
b = S.synth_batch(opt, B, 60, seed=1)
feats = [f.to(dev) for f in b["feats"]]; cat = b["category"].to(dev)
tr = Translator(m, dict(m.opt), device=dev)
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
