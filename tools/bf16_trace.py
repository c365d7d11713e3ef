"""Where does a workgroup of the bf16 matrix-core GEMM spend its cycles?  (tuning aid; needs `make -C .../csrc trace`)
Wave 0 of every workgroup accumulates shader-clock cycles per k-loop phase (csrc/gemm_bf16.hpp, NACF_BF16_TRACE):
compute | barrier | wait for the staged global loads | convert + LDS stores + next loads | barrier.
usage: NACF_GEMM_MODE=bf16x3 python tools/bf16_trace.py M:N:K [tile] [--images]     (forward GEMM only)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NACF_HIP_LIB", os.path.join(ROOT, "non-autoregressive-video-captioning_amd", "libnacf_hip_trace.so"))
import numpy as np
import torch
import nacf_amd  # noqa: F401
from nacf_amd.runtime import lib as L, ops

dev = torch.device("cuda:0")
M, N, K = (int(v) for v in sys.argv[1].split(":"))
args = sys.argv[2:]
images = "--images" in args
args = [a for a in args if not a.startswith("--")]
if args:
    os.environ["NACF_GEMM_TILE"] = args[0]
g = torch.Generator().manual_seed(0)
r = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).to(dev)
x, w, y = r(M, K), r(N, K), torch.empty(M, ops.vocab_ld(N), device=dev)[:, :N]
imgs = None
if images:
    imgs = ops.WeightImages(w.reshape(-1), [(0, N, K, False)], ops.gemm_mode())
    imgs.refresh()
f = lambda: ops.linear_fwd(x, w, y, None)
L.load()
raw = ctypes.CDLL(L.LIB_PATH)
raw.nacf_debug_bf16_trace.argtypes = [ctypes.c_void_p]
for _ in range(5):
    f()
torch.cuda.synchronize()
buf = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
assert raw.nacf_debug_bf16_trace(ctypes.c_void_p(buf.data_ptr())) == 0
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); f(); b.record()
torch.cuda.synchronize()
assert raw.nacf_debug_bf16_trace(ctypes.c_void_p(0)) == 0
print(L.load().nacf_gemm_last_kernel().decode())
raw16 = buf.cpu().numpy().reshape(-1, 16)
raw16 = raw16[raw16[:, 5] != 0]
raw8 = raw16[:, :8]
sub = raw8[:, 7].astype(np.uint64)
t = raw8.astype(np.float64)
nk = t[:, 6].max()
print("kernel %.1f us (with stamps), %d workgroups, %d k-tiles each" % (a.elapsed_time(b) * 1e3, len(t), nk))
names = ["compute", "barrier after compute", "wait global loads", "convert + LDS stores + next loads", "barrier after stores"]
tot = t[:, 5]
print("whole workgroup: p50 %.0f cycles (p10 %.0f, p90 %.0f); per k-tile %.0f" % (np.percentile(tot, 50), np.percentile(tot, 10), np.percentile(tot, 90), np.percentile(tot, 50) / nk))
for i, n in enumerate(names):
    v = t[:, i] / nk
    print("  %-34s per k-tile: p50 %6.0f  p10 %6.0f  p90 %6.0f   (%4.1f %% of the workgroup's life)" % (n, np.percentile(v, 50), np.percentile(v, 10), np.percentile(v, 90), 100 * t[:, i].sum() / tot.sum()))
print("  outside the k-loop (prologue + epilogue): %4.1f %%" % (100 * (1 - t[:, :5].sum() / tot.sum())))
if int(sub.max()) > 0:      # STAGES == 3 kernels: the staging phase in three parts
    parts = [(sub >> np.uint64(42)) & np.uint64((1 << 21) - 1), (sub >> np.uint64(21)) & np.uint64((1 << 21) - 1), sub & np.uint64((1 << 21) - 1)]
    for n, v in zip(["Q split + LDS stores (drained)", "Q load issue", "P LDS stores (drained)"], parts):
        print("    staging: %-32s per k-tile p50 %6.0f" % (n, np.percentile(v.astype(np.float64) / nk, 50)))
# ---- timeline: how many workgroups share a CU, and what happens outside the k-loop
t0 = raw16[:, 8].min()
beg, lend, end = (raw16[:, 8] - t0).astype(np.float64), (raw16[:, 9] - t0).astype(np.float64), (raw16[:, 10] - t0).astype(np.float64)
hw = raw16[:, 11].astype(np.uint64)
hwid, xcc = hw & np.uint64(0xffffffff), (hw >> np.uint64(32)) & np.uint64(0xf)
cu = ((xcc << np.uint64(16)) | (((hwid >> np.uint64(13)) & np.uint64(7)) << np.uint64(8)) | (((hwid >> np.uint64(12)) & np.uint64(1)) << np.uint64(4)) | ((hwid >> np.uint64(8)) & np.uint64(15))).astype(np.int64)
# the shader clock counters of the 8 XCDs have different origins: align each XCD to its own first workgroup
for x in set(xcc.tolist()):
    sel = xcc == x
    o = beg[sel].min()
    beg[sel] -= o; lend[sel] -= o; end[sel] -= o
issued = (raw16[:, 12] - t0).astype(np.float64)
for x in set(xcc.tolist()):
    sel = xcc == x
    issued[sel] -= (raw16[sel, 8] - t0).astype(np.float64).min()
span = end.max()
order = np.argsort(beg)
nq = 4
print("  epilogue by start-time quartile: " + "  ".join("q%d: issue %.0f + drain %.0f" % (i, np.median((issued - lend)[order[i * len(order) // nq:(i + 1) * len(order) // nq]]),
      np.median((end - issued)[order[i * len(order) // nq:(i + 1) * len(order) // nq]])) for i in range(nq)))
print("timeline: kernel span %.0f cycles (%.1f us at 2.1 GHz); %d distinct CUs; epilogue p50 %.0f cycles (p90 %.0f); prologue+loop p50 %.0f"
      % (span, span / 2.1e3, len(set(cu.tolist())), np.percentile(end - lend, 50), np.percentile(end - lend, 90), np.percentile(lend - beg, 50)))
print("  resident workgroups per CU, time-averaged: %.2f   (sum of lifetimes / (CUs * span))" % ((end - beg).sum() / (len(set(cu.tolist())) * span)))
starts = np.sort(beg)
q = [0, 10, 25, 50, 75, 90, 100]
print("  workgroup start times (cycles): " + "  ".join("p%d %.0f" % (x, np.percentile(starts, x)) for x in q))
print("  workgroup end   times (cycles): " + "  ".join("p%d %.0f" % (x, np.percentile(end, x)) for x in q))
per = {}
for c, b0, e0 in zip(cu.tolist(), beg.tolist(), end.tolist()):
    per.setdefault(c, []).append((b0, e0))
ncu = np.array([len(v) for v in per.values()])
print("  workgroups per CU: min %d  median %d  max %d" % (ncu.min(), np.median(ncu), ncu.max()))
c0 = sorted(per.items())[0]
print("  one CU's workgroups (start, end): " + "  ".join("(%.0f, %.0f)" % x for x in sorted(c0[1])))
if imgs is not None:
    imgs.close()
