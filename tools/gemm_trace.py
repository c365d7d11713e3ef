"""Where does a GEMM workgroup spend its cycles?  (tuning aid; needs `make -C .../csrc trace`)
Wave 0 of every workgroup stamps the shader clock at entry / after the prologue / after the k-loop / after the
epilogue; this prints the distribution, the per-k-tile cost, and how the workgroups were spread over CUs and time.
usage: NACF_HIP_LIB=.../libnacf_hip_trace.so python tools/gemm_trace.py kind:M:N:K [tile]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NACF_HIP_LIB", os.path.join(ROOT, "non-autoregressive-video-captioning_amd", "libnacf_hip_trace.so"))
import numpy as np
import torch
import nacf_amd  # noqa: F401
from nacf_amd.runtime import lib as L, ops

dev = torch.device("cuda:0")
kind, M, N, K = (int(v) for v in sys.argv[1].split(":"))
if len(sys.argv) > 2:
    os.environ["NACF_GEMM_TILE"] = sys.argv[2]
g = torch.Generator().manual_seed(0)
r = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).to(dev)
if kind == 0:
    x, w, y = r(M, K), r(N, K), torch.empty(M, ops.vocab_ld(N), device=dev)[:, :N]
    f = lambda: ops.linear_fwd(x, w, y, ops.Epi(bias=r(N)))
elif kind == 1:
    dz, w, dx = r(M, ops.vocab_ld(N))[:, :N], r(N, K), torch.empty(M, K, device=dev)
    f = lambda: ops.linear_bwd_data(dz, w, dx)
else:
    dz, x, dw = r(M, ops.vocab_ld(N))[:, :N], r(M, K), torch.empty(N, K, device=dev)
    f = lambda: ops.linear_bwd_weight(dz, x, dw, None, beta=0.0)
h = L.load()
raw = ctypes.CDLL(L.LIB_PATH)
raw.nacf_debug_gemm_trace.argtypes = [ctypes.c_void_p]
for _ in range(5):
    f()
torch.cuda.synchronize()
buf = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
assert raw.nacf_debug_gemm_trace(ctypes.c_void_p(buf.data_ptr())) == 0
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); f(); b.record()
torch.cuda.synchronize()
assert raw.nacf_debug_gemm_trace(ctypes.c_void_p(0)) == 0
t = buf.cpu().numpy().reshape(-1, 8)
t = t[t[:, 0] != 0]
n = len(t)
t0, t1, t2, t3, hw, nk, wall, wall_end = (t[:, i].astype(np.int64) for i in range(8))
ok = t3 > 0
print("kernel %.1f us, %d workgroups traced (%d reached the epilogue stamp), k-tiles/workgroup %d" % (a.elapsed_time(b) * 1e3, n, int(ok.sum()), int(nk.max())))
base = t0.min()
span = (t3[ok].max() - base)
pro, loop, epi = (t1 - t0)[ok], (t2 - t1)[ok], (t3 - t2)[ok]
q = lambda v: "min %6d  p50 %6d  p90 %6d  max %6d" % (v.min(), np.percentile(v, 50), np.percentile(v, 90), v.max())
print("span of all stamps: %d cycles" % span)
print("prologue  cycles:", q(pro))
print("k-loop    cycles:", q(loop), "  per k-tile p50 %.0f" % (np.percentile(loop, 50) / max(1, nk.max())))
print("epilogue  cycles:", q(epi))
print("total/wg  cycles:", q((t3 - t0)[ok]))
frac = lambda v: 100.0 * v.sum() / (pro.sum() + loop.sum() + epi.sum())
dw_ = (wall_end - wall)[ok].astype(np.float64)
good = dw_ > 0
if good.any():   # s_memtime ticks per 100 MHz wall tick between the post-fill and the final stamp = the shader clock there
    mhz = ((t3 - t1)[ok][good] / dw_[good]) * 100.0
    print("shader clock while the workgroups ran (s_memtime / wall_clock64): p10 %.0f  p50 %.0f  p90 %.0f MHz" % tuple(np.percentile(mhz, [10, 50, 90])))
print("share of workgroup time: prologue %.1f%%  k-loop %.1f%%  epilogue %.1f%%" % (frac(pro), frac(loop), frac(epi)))
# placement: HW_ID bits (gfx9): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5) | ((hw >> 32) << 8)
start = (t0 - base)
per_cu = {}
for c, s_, e_ in zip(cu[ok], start[ok], (t3 - base)[ok]):
    per_cu.setdefault(int(c), []).append((int(s_), int(e_)))
cnt = np.array([len(v) for v in per_cu.values()])
print("CUs used %d, workgroups per CU: min %d p50 %d max %d" % (len(per_cu), cnt.min(), np.median(cnt), cnt.max()))
busy_end = np.array([max(e for _, e in v) for v in per_cu.values()])
print("CU finish time (cycles from first start): min %d p50 %d max %d" % (busy_end.min(), np.median(busy_end), busy_end.max()))
late = np.sort(start[ok])
print("workgroup start times: p10 %d p50 %d p90 %d max %d" % tuple(np.percentile(late, [10, 50, 90, 100]).astype(int)))
# MFMA-bound time of the loop: per k-tile each wave issues TM*TN*4 MFMAs of 32 cycles (fp32 16x16x4 = 8 passes)
