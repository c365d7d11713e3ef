"""Loader-only throughput of the pinned-host placement: zero-copy gather vs per-clip DMA (tuning aid, GPU box only).
usage: python tools/loader_probe.py [n_frames]"""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nacf_amd
from nacf_amd.data import CaptionTable, FeatureShard, ShardLoader, write_feature_shard
dev = torch.device("cuda:0")
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N, T, D, B, L, V = 1024, 60, 2048, 128, 20, 1000
tmp = tempfile.mkdtemp()
try:
    rs = np.random.RandomState(0)
    shards = []
    for m in "mi":
        p = os.path.join(tmp, m + ".nacf"); write_feature_shard(p, rs.standard_normal((N, T, D)).astype(np.float32)); shards.append(FeatureShard(p))
    caps = {"video%d" % v: [[2] + rs.randint(6, V, size=8).tolist() + [3]] for v in range(N)}
    info = dict(itow={i: "w%d" % i for i in range(V)}, itop=None, itoc={v: 0 for v in range(N)}, length_info=None)
    opt = nacf_amd.opts.make_opt("NAB", "MSRVTT", with_category=True, max_len=L, vocab_size=V, n_frames=nf)
    table, vids = CaptionTable.from_corpus(caps, None, info, list(range(N)), opt, "train")
    for zc in (True, False):
        ld = ShardLoader(shards, table, vids, dict(opt, loader_zero_copy=zc), batch_size=B, device=dev, seed=1, placement="host", drop_last=True)
        for _ in ld: pass
        torch.cuda.synchronize(); t = time.perf_counter(); n = 0
        for _ in range(3):
            for b in ld: n += 1
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        print("n_frames %d  %s: %.2f ms/batch  %.0f videos/s" % (nf, "zero-copy gather" if zc else "per-clip DMA", dt / n * 1e3, n * B / dt))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
