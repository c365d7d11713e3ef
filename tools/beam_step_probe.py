"""nacf_beam_step alone (B = 256 instances, beam 5, V = 10547): HIP-event time per launch for near-uniform log-probs (random
init: every candidate is a contender) and for peaked ones."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nacf_amd  # noqa
from nacf_amd.runtime import ops
dev = torch.device("cuda:0")
B, n_bm, V, max_len, t = 256, 5, 10547, 20, 6
ld = ops.vocab_ld(V)
for name, scale in (("near-uniform", 1e-3), ("peaked", 5.0)):
    g = torch.Generator().manual_seed(0)
    lp = torch.log_softmax(torch.randn(B * n_bm, ld, generator=g)[:, :V] * scale, -1)
    buf = torch.zeros(B * n_bm, ld); buf[:, :V] = lp
    buf = buf.to(dev)
    seqs0 = torch.randint(6, V, (B, n_bm, max_len), generator=g).to(dev)
    ts = []
    for it in range(30):
        seqs = seqs0.clone(); scores = torch.rand(B, n_bm, generator=g).to(dev)
        fs, fl, ft = torch.zeros(B, n_bm, device=dev), torch.zeros(B, n_bm, dtype=torch.int32, device=dev), torch.zeros(B, n_bm, max_len, dtype=torch.int64, device=dev)
        fc, done, na = torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.beam_step(buf[:, :V], V, t, max_len, n_bm, seqs, scores, fs, fl, ft, fc, done, na)
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    print("%-13s median %.1f us (min %.1f)  [beam_step + count_active launches]" % (name, ts[len(ts) // 2], ts[0]))
