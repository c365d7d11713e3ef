import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nacf_amd
from nacf_amd.runtime import ops
dev = torch.device("cuda:0")
M, N, K = 5120, 2048, 512
x = torch.rand(M, K, device=dev); w = torch.rand(N, K, device=dev); y = torch.zeros(M, N, device=dev)
tok = (torch.rand(M, device=dev) < 0.575).long()
rs = ops.rowset_build(tokens=tok)
print("live", int(rs.count), "of", M)
def t(f, n=20):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
print("dense fwd ms", t(lambda: ops.linear_fwd(x, w, y, None)))
print("rows  fwd ms", t(lambda: ops.linear_fwd(x, w, y, None, rows=rs)))
dz = torch.rand(M, N, device=dev); dx = torch.zeros(M, K, device=dev); dw = torch.zeros(N, K, device=dev)
print("dense dX ms", t(lambda: ops.linear_bwd_data(dz, w, dx)))
print("rows  dX ms", t(lambda: ops.linear_bwd_data(dz, w, dx, rows=rs)))
print("dense dW ms", t(lambda: ops.linear_bwd_weight(dz, x, dw, None, beta=0.0)))
print("rows  dW ms", t(lambda: ops.linear_bwd_weight(dz, x, dw, None, beta=0.0, rows=rs)))
# model-level: is the decoder using row sets?
opt = nacf_amd.opts.make_opt("NACF", "MSRVTT", with_category=True, max_len=20, vocab_size=10547, fused_loss=True)
m = nacf_amd.get_model(opt).to(dev).train()
dec = m.decoder.bert
print("pack_rows", dec.pack_rows)
