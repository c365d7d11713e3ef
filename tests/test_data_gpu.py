"""GPU: SURVEY 8f row 1 -- device-side batch construction against the data oracle / the reference fixture, and the
shard loader end to end (resident and streaming modes deliver the same batches; a train step runs on them)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import nacf_data_oracle as D
from util import load_gold

pytestmark = pytest.mark.gpu


def _ops():
    import nacf_amd  # noqa: F401
    from nacf_amd.runtime import ops
    return ops


def _batch_arrays(g, dev):
    n = len(g["cap_len"])
    Lc = int(g["cap_len"].max())
    caps = np.zeros((n, Lc), dtype=np.int32)
    tags = np.zeros((n, Lc), dtype=np.int32)
    for i, k in enumerate(g["cap_len"]):
        caps[i, :k] = g["caps"][i, :k]
        tags[i, :k] = g["poss"][i, :k]
    up = lambda a, dt: torch.from_numpy(a).to(dev, dtype=dt)
    return (up(caps, torch.int32), up(g["cap_len"].astype(np.int32), torch.int32), up(tags, torch.int32),
            up(g["demanded"].astype(np.uint8), torch.uint8), up(g["is_be"].astype(np.uint8), torch.uint8))


@pytest.mark.parametrize("dt,vw,max_len", [("NARFormer", True, 10), ("NARFormer", False, 30), ("ARFormer", True, 10),
                                           ("ARFormer", False, 30)])
def test_build_targets_deterministic_parts_match_reference(dev, dt, vw, max_len):
    """everything that does not depend on a random draw is bit-exact vs the reference fixture: eval-mode MLM pairs,
    AR pairs in both modes, visual-word pairs"""
    ops = _ops()
    g = load_gold("tiny_data")
    caps, lens, tags, dem, be = _batch_arrays(g, dev)
    nar = dt == "NARFormer"
    rng = ops.RngState(5, dev)
    for mode in ("train", "validate"):
        key = "%s.%s.%d.%s" % (dt, "vw" if vw else "plain", max_len, mode)
        out = ops.build_targets(caps, lens, tags, dem, be, max_len, nar, vw, mode == "train", (0.35, 0.9), salt=1, rng=rng)
        names = ["tokens_1", "labels_1"] if (vw and mode == "train") else []
        if not (nar and mode == "train"):
            names += ["tokens", "labels"]
        assert names or nar
        for k in names:
            assert torch.equal(out[k].cpu(), torch.from_numpy(g[key + "." + k])), (key, k)


def test_build_targets_training_masks(dev):
    """training masks are random: check the reference's invariants per caption (dataloader.py:346-380) and that the
    number of masks and their positions are spread as the uniform draws prescribe"""
    ops = _ops()
    g = load_gold("tiny_data")
    caps, lens, tags, dem, be = _batch_arrays(g, dev)
    max_len, beta = 30, (0.35, 0.9)
    rng = ops.RngState(9, dev)
    counts = {i: [] for i in range(len(lens))}
    pos_hist = np.zeros(20)
    prev = None
    for it in range(400):
        out = ops.build_targets(caps, lens, None, None, None, max_len, True, False, True, beta, salt=3, rng=rng)
        rng.advance()
        tok, lab = out["tokens"].cpu().numpy(), out["labels"].cpu().numpy()
        if prev is not None and it == 1:
            assert not np.array_equal(prev, tok)                           # a new step draws new masks
        prev = tok
        for i, L in enumerate(g["cap_len"]):
            sent = g["caps"][i, 1:L - 1]
            n = len(sent)
            m = min(n, max_len)
            masked = tok[i, :m] == D.MASK
            assert np.array_equal(tok[i, :m][~masked], sent[:m][~masked])           # unmasked slots keep their word
            assert np.array_equal(lab[i, :m][masked], sent[:m][masked])             # masked slots carry the label
            assert (lab[i, :m][~masked] == D.PAD).all() and (tok[i, m:] == D.PAD).all() and (lab[i, m:] == D.PAD).all()
            rg = D.mlm_num_masks_range(n, beta)
            if rg is None:
                assert masked.sum() == 0
            elif n <= max_len:
                assert rg[0] <= masked.sum() < rg[1], (n, masked.sum(), rg)
                counts[i].append(int(masked.sum()))
            if n == 20:
                pos_hist += masked[:20]
    for i, L in enumerate(g["cap_len"]):                                  # every admissible count shows up
        rg = D.mlm_num_masks_range(int(L) - 2, beta)
        if rg is not None and int(L) - 2 <= max_len:
            assert set(counts[i]) == set(range(rg[0], rg[1])), (i, sorted(set(counts[i])), rg)
    assert pos_hist.min() > 0.75 * pos_hist.mean() and pos_hist.max() < 1.25 * pos_hist.mean()   # uniform positions
    # replay: same {seed, step} -> same masks
    r1, r2 = ops.RngState(9, dev), ops.RngState(9, dev)
    a = ops.build_targets(caps, lens, None, None, None, max_len, True, False, True, beta, salt=3, rng=r1)
    b = ops.build_targets(caps, lens, None, None, None, max_len, True, False, True, beta, salt=3, rng=r2)
    assert torch.equal(a["tokens"], b["tokens"]) and torch.equal(a["labels"], b["labels"])


def test_sample_frames(dev):
    ops = _ops()
    g = load_gold("tiny_data")
    rs = np.random.RandomState(0)
    N, T, Dm = 7, 60, 24
    src = torch.from_numpy(rs.rand(N, T, Dm).astype(np.float32)).to(dev)
    lens = torch.tensor([60, 60, 28, 9, 5, 60, 8], dtype=torch.int32, device=dev)
    video = torch.tensor([5, 2, 3, 4, 0, 6, 2, 1], dtype=torch.int32, device=dev)
    for n in (8, 60):
        out = torch.empty(len(video), n, Dm, device=dev)
        ids = torch.empty(len(video), n, dtype=torch.int32, device=dev)
        ops.sample_frames(src, video, lens, n, 0, out, frame_ids=ids)
        for b, v in enumerate(video.tolist()):
            S = int(lens[v])
            want = D.select_frames(S, n, n, 1, "equally_sampling")
            assert ids[b].tolist() == want, (n, S)
            assert torch.equal(out[b], src[v, torch.tensor(want, device=dev)])
    for total, n, ref in json.loads(str(g["frames_json"])):                # the reference's own frame-id cases
        s2 = torch.zeros(1, total, 4, device=dev)
        ids = torch.empty(1, n, dtype=torch.int32, device=dev)
        ops.sample_frames(s2, None, None, n, 0, torch.empty(1, n, 4, device=dev), frame_ids=ids)
        assert ids[0].tolist() == ref
    # segment_random: one uniform draw inside each segment; different steps differ; replay is exact
    rng = ops.RngState(4, dev)
    bound = D.frame_bounds(60, 8)
    seen = [set() for _ in range(8)]
    for _ in range(200):
        ids = torch.empty(2, 8, dtype=torch.int32, device=dev)
        ops.sample_frames(src, torch.tensor([0, 1], dtype=torch.int32, device=dev), lens, 8, 1, torch.empty(2, 8, Dm, device=dev),
                          salt=2, rng=rng, frame_ids=ids)
        rng.advance()
        for i, f in enumerate(ids[0].tolist()):
            assert bound[i] <= f < bound[i + 1]
            seen[i].add(f)
    assert all(len(s) == bound[i + 1] - bound[i] for i, s in enumerate(seen))
    # all_random (dataloader.py:25-26,37): n distinct frames of the clip, ascending; every frame is drawn about equally
    # often (n / S of the draws); a replay of the same {seed, step} is exact; a short clip is stretched as in the other modes
    rng = ops.RngState(9, dev)
    vid = torch.tensor([0, 2, 3], dtype=torch.int32, device=dev)          # clip lengths 60, 28, 9
    count = np.zeros((2, 60))
    trials = 600
    for t in range(trials):
        ids = torch.empty(3, 8, dtype=torch.int32, device=dev)
        out = torch.empty(3, 8, Dm, device=dev)
        ops.sample_frames(src, vid, lens, 8, 2, out, salt=3, rng=rng, frame_ids=ids)
        if t == 0:
            again = torch.empty_like(ids)
            ops.sample_frames(src, vid, lens, 8, 2, torch.empty_like(out), salt=3, rng=rng, frame_ids=again)
            assert torch.equal(ids, again)
        rng.advance()
        for b, S in enumerate((60, 28, 9)):
            f = ids[b].tolist()
            assert f == sorted(set(f)) and len(f) == 8 and 0 <= f[0] and f[-1] < S, (b, f)
            assert torch.equal(out[b], src[int(vid[b]), torch.tensor(f, device=dev)])
            if b < 2:
                count[b, f] += 1
    for b, S in enumerate((60, 28)):
        expect = trials * 8 / S
        assert count[b, S:].sum() == 0 and np.abs(count[b, :S] - expect).max() < 5 * np.sqrt(expect), (S, count[b, :S])
    ids = torch.empty(1, 8, dtype=torch.int32, device=dev)
    ops.sample_frames(src, torch.tensor([4], dtype=torch.int32, device=dev), lens, 8, 2, torch.empty(1, 8, Dm, device=dev), salt=3,
                      rng=rng, frame_ids=ids)
    assert ids[0].tolist() == D.select_frames(5, 8, 8, 1, "equally_sampling")      # 5 frames < 8: resampling (:20-21,305)


def _make_dataset(tmp, n_videos=12, T=60, Dm=32, seed=0):
    from nacf_amd.data import CaptionTable, FeatureShard, write_feature_shard
    rs = np.random.RandomState(seed)
    paths = []
    for m in "mi":
        p = os.path.join(tmp, "feats_%s.nacf" % m)
        write_feature_shard(p, rs.rand(n_videos, T, Dm).astype(np.float32), video_ids=np.arange(n_videos)[::-1].copy())
        paths.append(p)
    caps, tags, li = {}, {}, {}
    for v in range(n_videos):
        caps["video%d" % v] = [[D.BOS] + rs.randint(6, 101, size=rs.randint(3, 9)).tolist() + [D.EOS] for _ in range(3)]
        tags["video%d" % v] = [[2] + rs.randint(6, 12, size=len(c) - 2).tolist() + [3] for c in caps["video%d" % v]]
        h = [0] * 12
        for c in caps["video%d" % v]:
            h[len(c) - 2] += 1
        li["video%d" % v] = h
    info = dict(itow={i: ("is" if i == 6 else "w%d" % i) for i in range(101)},
                itop={i: t for i, t in enumerate(["<pad>", "<unk>", "<bos>", "<eos>", "<mask>", "<vis>", "NOUN", "VERB", "DET", "ADJ", "ADP", "PRON"])},
                itoc={v: v % 20 for v in range(n_videos)}, length_info=li)
    return [FeatureShard(p) for p in paths], caps, tags, info, CaptionTable


def test_shard_loader_resident_equals_streaming_and_feeds_a_train_step(dev, tmp_path):
    import nacf_amd
    from nacf_amd.data import ShardLoader
    from nacf_amd.misc.crit import get_criterion
    from nacf_amd.misc.optim import get_optimizer
    from nacf_amd import synthetic as S
    shards, caps, tags, info, CaptionTable = _make_dataset(str(tmp_path))
    opt = nacf_amd.opts.make_opt("NACF", "MSRVTT", with_category=True, dim_hidden=64, num_attention_heads=4,
                                 intermediate_size=128, dim_i=32, dim_m=32, max_len=10, hidden_dropout_prob=0.0,
                                 encoder_dropout=0.0, vocab_size=101, fused_loss=True, n_frames=8)
    table, vids = CaptionTable.from_corpus(caps, tags, info, list(range(12)), opt, "train")
    batches = {}
    for placement in ("hbm", "host", "host_dma", "host_kernel", "mmap"):
        # whole clips out of pinned memory: one DMA per clip | the copy kernel (nacf_gather_clips_zc, the default)
        o = dict(opt, loader_zero_copy=False, loader_clip_copy="dma" if placement == "host_dma" else "kernel") if placement.startswith("host_") else opt
        ld = ShardLoader(shards, table, vids, o, batch_size=8, device=dev, mode="train", seed=3, placement=placement[:4])
        assert len(ld) == (len(table) + 7) // 8 and ld.placement == placement[:4]
        assert ld.zero_copy == (placement == "host")          # 8 of 60 frames: the kernel gathers straight from pinned RAM
        batches[placement] = list(ld)
    batches[True] = batches["hbm"]
    for other in ("host", "host_dma", "host_kernel", "mmap"):         # the placements deliver identical batches
        for a, b in zip(batches["hbm"], batches[other]):
            assert set(a) == set(b)
            for k in a:
                xs, ys = (a[k], b[k]) if isinstance(a[k], list) else ([a[k]], [b[k]])
                assert all(torch.equal(torch.as_tensor(x), torch.as_tensor(y)) for x, y in zip(xs, ys)), (other, k)
    first = batches[True][0]
    n = first["tokens"].shape[0]
    assert first["feats"][0].shape == (n, 8, 32) and first["tokens"].shape == (n, 10) and first["category"].shape == (n, 1)
    assert torch.allclose(first["length_target"].sum(1), torch.ones(n, device=dev))
    # every feature row is a row of the right video's clip, inside the right segment
    idx = first["sample_index"].cpu().numpy()
    bound = D.frame_bounds(60, 8)
    for b in range(n):
        corpus_vid = int(vids[table.video[idx[b]]])
        clip = torch.from_numpy(np.array(shards[0].array[shards[0].row_of(np.array([corpus_vid]))[0]])).to(dev)
        for i in range(8):
            hit = (clip[bound[i]:bound[i + 1]] == first["feats"][0][b, i]).all(1)
            assert int(hit.sum()) >= 1
    # the batch drives a real training step
    model = nacf_amd.get_model(opt)
    model.load_state_dict(S.init_state_dict(opt, seed=0))
    model.to(dev).train()
    crit, optim = get_criterion(model.opt), get_optimizer(model.opt, model)
    optim.zero_grad()
    res = model(feats=first["feats"], tgt_tokens=[first["tokens_1"], first["tokens"]], category=first["category"])
    res["tgt_word_labels"] = [first["labels_1"], first["labels"]]
    res["tgt_length"] = first["length_target"]
    loss = crit.get_loss(res)
    loss.backward()
    optim.step()
    assert torch.isfinite(loss) and float(model.flat.grad.abs().max()) > 0
    # evaluation split: deterministic frames, everything masked
    tv, vv = CaptionTable.from_corpus(caps, tags, info, [1, 4], opt, "validate")
    ev = list(ShardLoader(shards, tv, vv, opt, batch_size=4, device=dev, mode="validate"))[0]
    assert len(tv) == 2 and "tokens_1" not in ev
    lens = torch.from_numpy(tv.cap_len.astype(np.int64) - 2).to(dev)
    assert torch.equal((ev["tokens"] == D.MASK).sum(1), lens.clamp(max=10))


@pytest.mark.parametrize("random_type", ["all_random", "segment_random"])
def test_shard_loader_shared_frame_ids_load_feats_type_0(dev, tmp_path, random_type):
    """load_feats_type 0 (dataloader.py:225-229,297-298): ONE frame-id draw per sample, shared by the modalities; with
    'all_random' the ids are distinct and ascending; every placement delivers the same batch"""
    import nacf_amd
    from nacf_amd.data import ShardLoader
    shards, caps, tags, info, CaptionTable = _make_dataset(str(tmp_path))
    opt = nacf_amd.opts.make_opt("NACF", "MSRVTT", with_category=True, dim_hidden=64, num_attention_heads=4,
                                 intermediate_size=128, dim_i=32, dim_m=32, max_len=10, vocab_size=101, n_frames=8,
                                 load_feats_type=0, random_type=random_type, n_total_frames=60)
    table, vids = CaptionTable.from_corpus(caps, tags, info, list(range(12)), opt, "train")
    got = {}
    for placement in ("hbm", "host"):
        got[placement] = list(ShardLoader(shards, table, vids, opt, batch_size=8, device=dev, mode="train", seed=5, placement=placement))[0]
    for a, b in zip(got["hbm"]["feats"], got["host"]["feats"]):
        assert torch.equal(a, b)
    first = got["hbm"]
    idx = first["sample_index"].cpu().numpy()
    differ = 0
    for b in range(first["tokens"].shape[0]):
        corpus_vid = int(vids[table.video[idx[b]]])
        clips = [torch.from_numpy(np.array(s.array[s.row_of(np.array([corpus_vid]))[0]])).to(dev) for s in shards]
        ids = []
        for i in range(8):
            hit = (clips[0] == first["feats"][0][b, i]).all(1).nonzero().flatten().tolist()
            assert len(hit) == 1
            ids.append(hit[0])
            assert torch.equal(first["feats"][1][b, i], clips[1][hit[0]])          # the other modality took the SAME frame
        if random_type == "all_random":
            assert ids == sorted(set(ids))
        else:
            bound = D.frame_bounds(60, 8)
            assert all(bound[i] <= f < bound[i + 1] for i, f in enumerate(ids))
        differ += ids != D.select_frames(60, 8, 8, 1, "equally_sampling")
    assert differ > 0
    with pytest.raises(ValueError):
        ShardLoader(shards, table, vids, dict(opt, n_total_frames=40), batch_size=8, device=dev, mode="train")
