"""Per-shape timing of the fp32 MFMA GEMM entry points (tuning aid, GPU box only).
usage: python tools/gemm_bench.py [--iters 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import nacf_amd  # noqa: E402,F401
from nacf_amd.runtime import ops  # noqa: E402

# (label, kind, M, N, K): the GEMMs of one NACF train step at B=128 and of one decode pass
SHAPES = [
    ("enc_lin fwd", 0, 7680, 512, 2048), ("enc_hw fwd", 0, 7680, 1024, 512), ("qkv fwd", 0, 5120, 1536, 512),
    ("kvmem fwd", 0, 15360, 1024, 512), ("proj fwd", 0, 5120, 512, 512), ("ffn1 fwd", 0, 5120, 2048, 512),
    ("ffn2 fwd", 0, 5120, 512, 2048), ("vocab fwd", 0, 5120, 10547, 512), ("vocab dec", 0, 14592, 10547, 512),
    ("vocab dX", 1, 5120, 10547, 512), ("ffn2 dX", 1, 5120, 512, 2048), ("ffn1 dX", 1, 5120, 2048, 512),
    ("proj dX", 1, 5120, 512, 512), ("kvmem dX", 1, 15360, 1024, 512), ("hw dX", 1, 7680, 1024, 512),
    ("vocab dW", 2, 5120, 10547, 512), ("enc_lin dW", 2, 7680, 512, 2048), ("ffn1 dW", 2, 5120, 2048, 512),
    ("ffn2 dW", 2, 5120, 512, 2048), ("proj dW", 2, 5120, 512, 512), ("kvmem dW", 2, 15360, 1024, 512),
]


PAD = 0


def run(kind, M, N, K, iters, dev, images=False):
    g = torch.Generator(device="cpu").manual_seed(0)
    r = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).to(dev)
    imgs = None
    mode = ops.gemm_mode()
    if kind == 0:
        x, w, y = r(M, K + PAD)[:, :K], r(N, K + PAD)[:, :K], torch.empty(M, ops.vocab_ld(N), device=dev)[:, :N]
        if images and mode != 0 and PAD == 0:
            imgs = ops.WeightImages(w.reshape(-1), [(0, N, K, False)], mode)
            imgs.refresh()
        f = lambda: ops.linear_fwd(x, w, y, None)
    elif kind == 1:
        dz, w, dx = r(M, ops.vocab_ld(N))[:, :N], r(N, K), torch.empty(M, K, device=dev)
        if images and mode != 0:
            imgs = ops.WeightImages(w.reshape(-1), [(0, N, K, True)], mode)
            imgs.refresh()
        f = lambda: ops.linear_bwd_data(dz, w, dx)
    else:
        dz, x, dw = r(M, ops.vocab_ld(N))[:, :N], r(M, K), torch.empty(N, K, device=dev)
        f = lambda: ops.linear_bwd_weight(dz, x, dw, None, beta=0.0)
    for _ in range(3):
        f()
    # one event pair per launch, MEDIAN over the launches: a mean over a back-to-back run picks up one-off stalls of
    # tens of milliseconds (round 2's `proj fwd 3.934 ms` and the like: a code object or workspace touched for the first
    # time inside the timed run, another tenant of the box) as if they were the kernel's time
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    torch.cuda.synchronize()
    for a, b in evs:
        a.record()
        f()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    ms = ts[len(ts) // 2]
    if ts[-1] > 5 * ms:
        print("# %s %d,%d,%d: one launch of %d took %.3f ms (median %.3f): not the kernel" % (("fwd", "dX", "dW")[kind], M, N, K, iters, ts[-1], ms),
              file=sys.stderr)
    if imgs is not None:
        imgs.close()
    return ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--tiles", type=str, default="64,128", help="64 | 128 (NACF_GEMM_TILE), wide1 | wide2 (NACF_GEMM_WIDE) / dma1 | dma2 (NACF_DMA128) (exact mode + --images), auto")
    ap.add_argument("--shapes", type=str, default="", help="extra shapes 'kind:M:N:K,...' (kind 0 fwd, 1 dX, 2 dW)")
    ap.add_argument("--pad", type=int, default=0, help="extra floats of row pitch on the K-contiguous fwd operands")
    ap.add_argument("--modes", type=str, default="f32", help="NACF_GEMM_MODE values to compare: f32,bf16x3,bf16")
    ap.add_argument("--images", action="store_true", help="bf16 modes: weights come from pre-split images (as in the model)")
    args = ap.parse_args()
    global PAD
    PAD = args.pad
    dev = torch.device("cuda:0")
    tiles = [m + "/" + t for m in args.modes.split(",") for t in args.tiles.split(",")]
    print("%-12s %-22s " % ("gemm", "M,N,K") + " ".join("%13s %7s" % ("ms@" + t, "TF") for t in tiles))
    shapes = SHAPES
    if args.shapes:
        shapes = [("custom k%s" % t.split(":")[0],) + tuple(int(v) for v in t.split(":")) for t in args.shapes.split(",")]
    for label, kind, M, N, K in shapes:
        if args.only and args.only not in label:
            continue
        res = []
        for mt in tiles:
            mode_, tile_ = mt.split("/")
            os.environ["NACF_GEMM_MODE"] = mode_
            os.environ.pop("NACF_GEMM_TILE", None)
            os.environ.pop("NACF_GEMM_WIDE", None)
            os.environ.pop("NACF_DMA128", None)
            if tile_ in ("64", "128"):
                os.environ["NACF_GEMM_TILE"] = tile_             # (also keeps the wide and DMA-fed kernels out)
            elif tile_.startswith("wide"):                       # wide1 / wide2: csrc/gemm_bf16_wide.hpp, 64 / 128-row tiles
                os.environ["NACF_GEMM_WIDE"] = tile_[4:]
                os.environ["NACF_DMA128"] = "0"
            elif tile_.startswith("dma"):                        # dma1 / dma2: csrc/gemm_dma128.hpp, 64 / 128-row tiles
                os.environ["NACF_GEMM_WIDE"] = "0"
                os.environ["NACF_DMA128"] = tile_[3:]
            else:
                assert tile_ == "auto", tile_                    # the production heuristic
            res.append(run(kind, M, N, K, args.iters, dev, args.images))
        print("%-12s %-22s " % (label, "%d,%d,%d" % (M, N, K)) + " ".join("%13.3f %7.1f" % r for r in res))


if __name__ == "__main__":
    main()
