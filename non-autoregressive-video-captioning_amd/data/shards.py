"""On-disk formats.

The reference keeps one HDF5 dataset per video (`video<id>`: [<=60, 2048] fp32,
pretreatment/extract_image_feats_from_frames.py:54-57) and opens it per sample in `__getitem__`
(dataloader.py:132-144,263-315).  At 25k+ videos/s that access pattern cannot feed one MI355X, let alone eight,
so features are re-packed once into a **feature shard**: a 4 KiB header + ONE row-major fp32 array
[N, T, D] (clips shorter than T are zero-padded, their true length is kept), page-aligned so it can be
memory-mapped, read with large sequential I/O, pinned, or uploaded whole -- MSRVTT's 10 000 x 60 x 2048 rows are
4.9 GB per modality, a small corner of 288 GB of HBM.

Captions, POS tags, categories and length histograms (info_corpus pickle: prepare_corpora.py:38-60) become a
**caption table** of flat int32 arrays, one row per (video, caption) pair in the order the reference's
`_make_infoset` enumerates them (dataloader.py:146-199).
"""
import json
import os

import numpy as np

MAGIC = b"NACFSHD1"
HEADER_BYTES = 4096
BE_VERBS = ('is', 'are', 'was', 'were', 'be')      # dataloader.py:402


def write_feature_shard(path, feats, lengths=None, video_ids=None):
    """feats: float32 [N, T, D] (or an iterable of [t_i, D] arrays, zero-padded to the longest / `T`)."""
    if not isinstance(feats, np.ndarray):
        rows = [np.asarray(f, dtype=np.float32) for f in feats]
        T = max(r.shape[0] for r in rows)
        lengths = np.array([r.shape[0] for r in rows], dtype=np.int32)
        arr = np.zeros((len(rows), T, rows[0].shape[1]), dtype=np.float32)
        for i, r in enumerate(rows):
            arr[i, :r.shape[0]] = r
        feats = arr
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    N, T, D = feats.shape
    if lengths is None:
        lengths = np.full(N, T, dtype=np.int32)
    if video_ids is None:
        video_ids = np.arange(N, dtype=np.int64)
    meta = dict(N=N, T=T, D=D, dtype="float32", data_offset=HEADER_BYTES,
                lengths_offset=HEADER_BYTES + feats.nbytes, ids_offset=HEADER_BYTES + feats.nbytes + 4 * N)
    head = MAGIC + json.dumps(meta).encode()
    assert len(head) < HEADER_BYTES
    with open(path, "wb") as f:
        f.write(head.ljust(HEADER_BYTES, b"\0"))
        f.write(feats.tobytes())
        f.write(np.asarray(lengths, dtype=np.int32).tobytes())
        f.write(np.asarray(video_ids, dtype=np.int64).tobytes())


class FeatureShard:
    """Read side: `array` is a read-only memmap [N, T, D]; `lengths` the true clip lengths; `video_ids` the keys."""

    def __init__(self, path):
        with open(path, "rb") as f:
            head = f.read(HEADER_BYTES)
        if head[:len(MAGIC)] != MAGIC:
            raise ValueError("%s is not a NACF feature shard" % path)
        meta = json.loads(head[len(MAGIC):].rstrip(b"\0").decode())
        self.path, self.N, self.T, self.D = path, meta["N"], meta["T"], meta["D"]
        self.array = np.memmap(path, dtype=np.float32, mode="r", offset=meta["data_offset"], shape=(self.N, self.T, self.D))
        self.lengths = np.array(np.memmap(path, dtype=np.int32, mode="r", offset=meta["lengths_offset"], shape=(self.N,)))
        self.video_ids = np.array(np.memmap(path, dtype=np.int64, mode="r", offset=meta["ids_offset"], shape=(self.N,)))
        self.nbytes = self.N * self.T * self.D * 4

    def row_of(self, video_ids):
        """positions of the given video ids inside this shard"""
        order = np.argsort(self.video_ids)
        pos = np.searchsorted(self.video_ids[order], video_ids)
        rows = order[np.clip(pos, 0, self.N - 1)]
        if not np.array_equal(self.video_ids[rows], np.asarray(video_ids)):
            raise KeyError("video id missing from %s" % self.path)
        return rows


class CaptionTable:
    """Flat arrays for every (video, caption) sample, in `_make_infoset` order (dataloader.py:146-199)."""

    FIELDS = ("caps", "cap_len", "pos_tags", "video", "cap_id", "category", "length_target", "tag_demanded",
              "word_is_be")

    def __init__(self, **arrays):
        for k in self.FIELDS:
            setattr(self, k, arrays[k])

    def __len__(self):
        return int(self.caps.shape[0])

    @classmethod
    def from_corpus(cls, captions, pos_tags, info, video_indices, opt, mode, rng=None):
        """captions / pos_tags: {'video<i>': [[<bos> ... <eos>], ...]}; info: the 'info' dict of the corpus pickle
        (itow, itop, itoc, length_info); video_indices: the split (info['split'][mode]); rng: np.random.RandomState
        used when opt['n_caps_per_video'] > 0 in training (the reference draws from RandomState(opt['seed']))."""
        itow, itop, itoc = info["itow"], info.get("itop"), info.get("itoc")
        length_info = info.get("length_info")
        max_len = opt["max_len"]
        train = mode == "train"
        n_caps = opt.get("n_caps_per_video", 0) if train else (0 if opt.get("parallel_mlm", False) else 1)
        rows, lt_rows, cat_rows, vid_rows = [], [], [], []
        for ix in [int(v) for v in video_indices]:
            vid = "video%d" % ix
            caps = captions[vid]
            tags = pos_tags[vid] if pos_tags is not None else [None] * len(caps)
            assert len(caps) == len(tags)
            if length_info is None:
                lt = np.zeros(max_len)
            else:                                                        # dataloader.py:166-175
                lt = list(length_info[vid])[:max_len]
                lt += [0] * (max_len - len(lt))
                lt = np.array(lt) / sum(lt)
            if n_caps == 0:
                ids = list(range(len(caps)))
            elif n_caps == 1 and not train:
                ids = [0]
            else:
                ids = list(rng.choice(list(range(len(caps))), min(len(caps), n_caps), replace=False))
            for c in ids:
                rows.append((len(vid_rows), int(c), caps[c], tags[c]))
            vid_rows.append(ix)
            lt_rows.append(lt)
            cat_rows.append(itoc[ix] if itoc is not None else 0)
        Lc = max(len(r[2]) for r in rows)
        n = len(rows)
        cap_arr = np.zeros((n, Lc), dtype=np.int32)
        tag_arr = np.zeros((n, Lc), dtype=np.int32)
        for i, (_, _, c, t) in enumerate(rows):
            cap_arr[i, :len(c)] = c
            if t is not None:
                tag_arr[i, :len(t)] = t
        demand = set(opt.get("demand", ["VERB", "NOUN"]))
        n_tags = (max(itop) + 1) if itop else 1
        demanded = np.array([(itop is not None and itop.get(i) in demand) for i in range(n_tags)], dtype=np.uint8)
        n_words = max(itow) + 1
        is_be = np.array([itow.get(i) in BE_VERBS for i in range(n_words)], dtype=np.uint8)
        return cls(caps=cap_arr, cap_len=np.array([len(r[2]) for r in rows], dtype=np.int32), pos_tags=tag_arr,
                   video=np.array([r[0] for r in rows], dtype=np.int32), cap_id=np.array([r[1] for r in rows], dtype=np.int32),
                   category=np.array(cat_rows, dtype=np.int64), length_target=np.array(lt_rows, dtype=np.float32),
                   tag_demanded=demanded, word_is_be=is_be, ), np.array(vid_rows, dtype=np.int64)

    def save(self, path):
        np.savez_compressed(path, **{k: getattr(self, k) for k in self.FIELDS})

    @classmethod
    def load(cls, path):
        z = np.load(path)
        return cls(**{k: z[k] for k in cls.FIELDS})
