"""Batch iterator over feature shards + a caption table.

Replaces `DataLoader(VideoDataset(...), batch_size, shuffle)` of the reference (dataloader.py:40-425,
misc/run.py:92-96) with the same batch dictionary -- feats (one [B, n_frames, D] tensor per modality), tokens, labels,
tokens_1 / labels_1 (visual-word pass), length_target, category -- built ON THE DEVICE:

  * resident mode (default when the shards fit `hbm_budget_bytes`): the whole [N, T, D] arrays are uploaded once; a
    batch is one gather+frame-sampling launch per modality driven by a device index vector.  No PCIe traffic per step.
  * host mode (shards that exceed the HBM budget but fit pinned host memory): the shard is read once into pinned RAM.
    While the current batch trains, the next one is fetched on a side stream, one of two ways:
      - few frames per clip (n_frames <= T/3, e.g. the reference's 8 of 60): pinned memory is mapped into the device's
        address space and the frame-sampling kernel itself pulls exactly the sampled rows over PCIe (zero-copy gather:
        7.5x fewer bytes than copying clips, no host loop per clip; measured ~18 GB/s of kernel-issued PCIe reads);
      - most frames needed: the whole clips (T*D*4 contiguous bytes each) go to a device staging buffer, frame sampling then
        runs on the staged clips.  Default: a copy KERNEL reading the pinned shard over PCIe (`nacf_gather_clips_zc`, 24
        workgroups with 16 x 16 bytes per thread in flight): 34.0 k videos/s on the NACF step at 128 videos against 24.3 k
        for one asynchronous DMA per clip (`nacf_gather_clips_h2d`, opt['loader_clip_copy'] = 'dma': 256 copies of 491 KB
        per batch top out at ~26 GB/s); measured 8 / 16 / 24 / 32 / 64 / 128 workgroups: 22 / 34 / 34 / 33 / 29 / 27 k --
        more workgroups take compute units from the training step, fewer do not cover the link's latency.
  * mmap mode (larger than host memory): the batch's clips are copied from the memory-mapped shard into one of two
    pinned staging buffers by a few host threads and uploaded on the side stream (double buffering).
    In both streaming modes frame sampling then runs on the staged rows.
  * decoder inputs / labels (masked-LM pairs, visual-word targets, AR pairs) come from `nacf_build_targets`; the
    masks are drawn with the device Philox stream {seed, step}, so an epoch is reproducible from `seed`.

There is no CPU fallback: construction raises without the HIP library / a HIP device.
"""
import os

import numpy as np
import torch

from ..runtime import ops


class ShardLoader:
    def __init__(self, shards, table, video_index, opt, batch_size, device, mode="train", shuffle=None, seed=0,
                 resident=None, hbm_budget_bytes=64 << 30, drop_last=False, placement=None, host_budget_bytes=64 << 30,
                 rank=0, world=1):
        """shards: one FeatureShard per modality character of opt['modality'] (same video ids); table / video_index:
        CaptionTable.from_corpus(...) output (video_index = corpus ids of the table's video rows).
        rank / world: data-parallel sharding -- every rank draws the SAME permutation (same seed) and takes the
        rank-th slice of `batch_size` samples out of each global batch of world*batch_size; the ragged tail is dropped
        so that all ranks run the same number of steps."""
        self.rank, self.world = int(rank), int(world)
        assert 0 <= self.rank < self.world
        self._into = None
        self.shards, self.table, self.opt = list(shards), table, opt
        self.B, self.dev, self.mode = int(batch_size), torch.device(device), mode
        self.train = mode == "train"
        self.shuffle = self.train if shuffle is None else bool(shuffle)
        self.drop_last = drop_last or self.world > 1
        lft = opt.get("load_feats_type", 1)
        if lft not in (0, 1, 2):
            raise ValueError("nacf_amd: load_feats_type must be 0, 1 or 2 (dataloader.py:297-311)")
        # load_feats_type 0 (dataloader.py:225-229,297-298): ONE frame-id draw per sample out of opt['n_total_frames'],
        # shared by every modality and not adapted to the clip's own length
        self.shared_frames = lft == 0
        if self.shared_frames:
            nt = opt.get("n_total_frames", 60)
            if any(s.T != nt for s in shards):
                raise ValueError("nacf_amd: load_feats_type 0 draws frame ids out of n_total_frames = %d; the shards hold %s "
                                 "frames per clip" % (nt, [s.T for s in shards]))
        random_type = opt.get("random_type", "segment_random") if self.train else "equally_sampling"
        if random_type not in ("segment_random", "all_random", "equally_sampling"):
            raise ValueError("nacf_amd: random_type %s (dataloader.py:52)" % random_type)
        self.frame_mode = {"equally_sampling": 0, "segment_random": 1, "all_random": 2}[random_type]
        self.n_frames = [s.T if opt.get("load_feats_type", 1) == 2 else opt["n_frames"] for s in self.shards]
        self.rows = [s.row_of(video_index) for s in self.shards]          # table video row -> shard row
        total = sum(s.nbytes for s in self.shards)
        if placement is None:        # where the shards live: "hbm" | "host" (pinned RAM) | "mmap" (page cache / disk)
            if resident is not None:
                placement = "hbm" if resident else "mmap"
            else:
                placement = "hbm" if total <= hbm_budget_bytes else ("host" if total <= host_budget_bytes else "mmap")
        if placement not in ("hbm", "host", "mmap"):
            raise ValueError("placement must be hbm | host | mmap")
        self.placement = placement
        self.resident = placement == "hbm"
        self.zero_copy = False
        self.rng = ops.RngState(seed, self.dev)
        self.gen = torch.Generator().manual_seed(seed)
        t = table
        up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev, dtype=dt)
        self.d_caps, self.d_len, self.d_tags = up(t.caps, torch.int32), up(t.cap_len, torch.int32), up(t.pos_tags, torch.int32)
        self.d_video, self.d_cat = up(t.video, torch.int64), up(t.category, torch.int64)
        self.d_lt = up(t.length_target, torch.float32)
        self.d_dem, self.d_be = up(t.tag_demanded, torch.uint8), up(t.word_is_be, torch.uint8)
        self.d_srclen = [up(s.lengths, torch.int32) for s in self.shards]
        if self.resident:
            self.d_feats = [self._upload(s) for s in self.shards]
            self.d_rows = [up(r, torch.int32) for r in self.rows]
        else:
            self.copy_stream = torch.cuda.Stream(device=self.dev, priority=int(os.environ.get('NACF_LOADER_STREAM_PRIORITY', '0')))
            self.zero_copy = False
            if placement == "host":
                self.host_feats = []
                for s in self.shards:
                    h = torch.empty(s.N, s.T, s.D, dtype=torch.float32).pin_memory()
                    np.copyto(h.numpy(), s.array)
                    self.host_feats.append(h)
                self.pinned = None
                zc = opt.get("loader_zero_copy")
                if zc is None and os.environ.get("NACF_LOADER_ZERO_COPY", "") != "":
                    zc = os.environ["NACF_LOADER_ZERO_COPY"] != "0"
                self.zero_copy = bool(zc) if zc is not None else all(3 * nf <= s.T for nf, s in zip(self.n_frames, self.shards))
                # per staging slot: the sampled batch, and a private {seed, step} so that the prefetch kernel draws
                # the frames of batch k with step k no matter when it runs relative to the compute stream
                self.sampled = [[torch.empty(self.B, nf, s.D, dtype=torch.float32, device=self.dev)
                                 for nf, s in zip(self.n_frames, self.shards)] for _ in range(2)]
                self.pref_rng = [ops.RngState(seed, self.dev) for _ in range(2)]
                self.pref_state = [torch.tensor([seed, 0], dtype=torch.int64).pin_memory() for _ in range(2)]
                self.pref_rows = [[torch.empty(self.B, dtype=torch.int32).pin_memory() for _ in self.shards] for _ in range(2)]
                self.pref_rows_dev = [[torch.empty(self.B, dtype=torch.int32, device=self.dev) for _ in self.shards]
                                      for _ in range(2)]
                self.n_built = 0
                # whole clips out of pinned memory: "kernel" = kernel-issued PCIe reads, "dma" = one hipMemcpyAsync per clip
                self.clip_copy = opt.get("loader_clip_copy", os.environ.get("NACF_LOADER_CLIP_COPY", "kernel"))
                if self.clip_copy not in ("kernel", "dma"):
                    raise ValueError("loader_clip_copy must be 'kernel' or 'dma', not %r" % (self.clip_copy,))
                # nacf_gather_clips_zc moves 16-byte packets: a shard whose clips are not a multiple of 16 bytes takes the per-clip DMA
                if self.clip_copy == "kernel" and any((int(s_.T) * int(s_.D) * 4) % 16 != 0 for s_ in self.shards):
                    self.clip_copy = "dma"
                self.clip_copy_wgs = int(opt.get("loader_clip_copy_wgs", os.environ.get("NACF_LOADER_CLIP_COPY_WGS", "24")))
            else:
                self.pinned = [[torch.empty(self.B, s.T, s.D, dtype=torch.float32).pin_memory() for s in self.shards]
                               for _ in range(2)]
            if placement == "mmap" or not self.zero_copy:
                self.staged = [[torch.empty(self.B, s.T, s.D, dtype=torch.float32, device=self.dev) for s in self.shards] for _ in range(2)]
                self.staged_len = [[torch.empty(self.B, dtype=torch.int32, device=self.dev) for _ in self.shards] for _ in range(2)]
                self.pinned_len = [[torch.empty(self.B, dtype=torch.int32).pin_memory() for _ in self.shards] for _ in range(2)]
            self._slot_event = [None, None]
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=int(opt.get("loader_threads", 32)))

    def _upload(self, shard, chunk_bytes=256 << 20):
        """whole shard -> HBM: large sequential reads of the memory-mapped file into two alternating pinned buffers,
        asynchronous H2D copies behind them"""
        dst = torch.empty(shard.N, shard.T, shard.D, dtype=torch.float32, device=self.dev)
        per = max(1, chunk_bytes // (shard.T * shard.D * 4))
        pins = [torch.empty(min(per, shard.N), shard.T, shard.D, dtype=torch.float32).pin_memory() for _ in range(2)]
        evs = [None, None]
        for k, lo in enumerate(range(0, shard.N, per)):
            hi, slot = min(shard.N, lo + per), k & 1
            if evs[slot] is not None:
                evs[slot].synchronize()
            np.copyto(pins[slot].numpy()[:hi - lo], shard.array[lo:hi])
            dst[lo:hi].copy_(pins[slot][:hi - lo], non_blocking=True)
            evs[slot] = torch.cuda.Event()
            evs[slot].record()
        torch.cuda.current_stream(self.dev).synchronize()
        return dst

    def __len__(self):
        n, g = len(self.table), self.B * self.world
        return n // g if self.drop_last else (n + g - 1) // g

    # ---- host side of streaming mode: gather the batch's clips into pinned memory, start the upload
    def _stage(self, slot, vids_host, batch_no=0):
        n = len(vids_host)
        prev = self._slot_event[slot]
        if prev is not None:
            prev.synchronize()              # the upload that last read this pinned slot has finished
        # the frame-sampling kernels that read this staging slot two batches ago were launched on the compute stream
        self.copy_stream.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.copy_stream):
            if self.placement == "host" and self.zero_copy:
                self.pref_state[slot][1] = batch_no
                self.pref_rng[slot].state.copy_(self.pref_state[slot], non_blocking=True)
            for m, s in enumerate(self.shards):
                rows = self.rows[m][vids_host]
                if self.placement == "host" and not self.zero_copy:    # whole clips, pinned RAM -> device staging
                    if self.clip_copy == "kernel":                      # kernel-issued PCIe reads (nacf_gather_clips_zc)
                        self.pref_rows[slot][m][:n] = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int32))
                        self.pref_rows_dev[slot][m][:n].copy_(self.pref_rows[slot][m][:n], non_blocking=True)
                        ops.gather_clips_zc(self.staged[slot][m], self.host_feats[m], self.pref_rows_dev[slot][m][:n], self.clip_copy_wgs)
                    else:                                               # one DMA per clip
                        ops.gather_clips_h2d(self.staged[slot][m], self.host_feats[m], rows)
                    self.pinned_len[slot][m][:n] = torch.from_numpy(s.lengths[rows])
                    self.staged_len[slot][m][:n].copy_(self.pinned_len[slot][m][:n], non_blocking=True)
                    continue
                if self.placement == "host":               # zero-copy gather: the kernel reads the pinned shard itself
                    self.pref_rows[slot][m][:n] = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int32))
                    self.pref_rows_dev[slot][m][:n].copy_(self.pref_rows[slot][m][:n], non_blocking=True)
                    ops.sample_frames(self.host_feats[m], self.pref_rows_dev[slot][m][:n],
                                      None if self.shared_frames else self.d_srclen[m],
                                      self.n_frames[m], self.frame_mode, self.sampled[slot][m][:n],
                                      salt=self._frame_salt(m), rng=self.pref_rng[slot])
                    continue
                buf = self.pinned[slot][m].numpy()
                # one memcpy per clip (T*D*4 bytes, contiguous in the shard) straight into pinned memory, spread over a
                # few host threads (numpy drops the GIL for plain copies); no intermediate gather buffer
                list(self._pool.map(lambda jr: np.copyto(buf[jr[0]], s.array[jr[1]]), enumerate(rows)))
                self.staged[slot][m][:n].copy_(self.pinned[slot][m][:n], non_blocking=True)
                self.pinned_len[slot][m][:n] = torch.from_numpy(s.lengths[rows])
                self.staged_len[slot][m][:n].copy_(self.pinned_len[slot][m][:n], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self._slot_event[slot] = ev
        return ev

    def _frame_salt(self, m):
        """the Philox salt of modality m's frame draw: one per modality, or one for all of them under load_feats_type 0
        (the same {seed, step, salt, sample} gives the same frame ids: dataloader.py:225-229)"""
        return 0x5EED0000 if self.shared_frames else 0x5EED0000 + m

    def bind_outputs(self, buffers):
        """Build every following batch straight INTO these tensors (a dict shaped like a batch, e.g. the step engine's
        static graph inputs) whenever their shapes fit: the hand-over to a captured training step then costs no copy.
        The tensors of a yielded batch are overwritten by the next one."""
        self._into = buffers

    def _slot(self, name, shape, dtype, index=None):
        t = (self._into or {}).get(name)
        if index is not None and isinstance(t, (list, tuple)):
            t = t[index] if index < len(t) else None
        if torch.is_tensor(t) and tuple(t.shape) == tuple(shape) and t.dtype == dtype and t.is_contiguous() and t.device == self.dev:
            return t
        return torch.empty(shape, dtype=dtype, device=self.dev)

    def _build(self, idx_dev, vids_dev, feats_src, idx_host=None):
        opt, n = self.opt, idx_dev.numel()
        batch = {"feats": []}
        for m, s in enumerate(self.shards):
            src, video, src_len = feats_src(m, vids_dev)
            out = self._slot("feats", (n, self.n_frames[m], s.D), torch.float32, m)
            if video is None and src_len is None:          # host placement: already sampled by the prefetch kernel
                batch["feats"].append(out.copy_(src))      # (the staging slot is overwritten two batches later)
                continue
            ops.sample_frames(src, video, None if self.shared_frames else src_len, self.n_frames[m], self.frame_mode, out,
                              salt=self._frame_salt(m), rng=self.rng)
            batch["feats"].append(out)
        caps, lens = self.d_caps.index_select(0, idx_dev), self.d_len.index_select(0, idx_dev)
        tags = self.d_tags.index_select(0, idx_dev)
        narformer = opt["decoding_type"] == "NARFormer"
        batch.update(ops.build_targets(caps, lens, tags, self.d_dem, self.d_be, opt["max_len"], narformer,
                                       opt.get("visual_word_generation", False), self.train, opt.get("beta", [0, 1]),
                                       salt=0x7A26E7, rng=self.rng, into=self._into))
        lt = self._slot("length_target", (n, self.d_lt.shape[1]), torch.float32)
        batch["length_target"] = torch.index_select(self.d_lt, 0, vids_dev, out=lt)
        cat = self._slot("category", (n, 1), torch.int64)
        torch.index_select(self.d_cat, 0, vids_dev, out=cat.view(n))
        batch["category"] = cat
        batch["sample_index"] = idx_dev
        batch["sample_index_host"] = idx_host       # numpy: lets the caller name the videos without a device sync
        self.rng.advance()
        return batch

    def __iter__(self):
        n = len(self.table)
        order = torch.randperm(n, generator=self.gen) if self.shuffle else torch.arange(n)
        nb = len(self)
        g, lo = self.B * self.world, self.B * self.rank
        chunks = [order[i * g + lo:min(n, i * g + lo + self.B)] for i in range(nb)]
        if self.resident:
            for ch in chunks:
                idx = ch.to(self.dev)
                vids = self.d_video.index_select(0, idx)
                yield self._build(idx, vids, lambda m, v: (self.d_feats[m], self.d_rows[m].index_select(0, v), self.d_srclen[m]),
                                  ch.numpy())
            return
        vid_host = self.table.video
        host = self.placement == "host" and self.zero_copy
        base = self.n_built if host else 0         # Philox step of a batch = number of batches built before it
        pending = self._stage(0, vid_host[chunks[0].numpy()], base) if nb else None
        for i, ch in enumerate(chunks):
            slot = i & 1
            ev = pending
            if i + 1 < nb:                                                 # start the next upload before this batch is used
                pending = self._stage(1 - slot, vid_host[chunks[i + 1].numpy()], base + i + 1)
            torch.cuda.current_stream(self.dev).wait_event(ev)
            idx = ch.to(self.dev)
            vids = self.d_video.index_select(0, idx)
            k = idx.numel()
            if host:
                self.n_built += 1
                yield self._build(idx, vids, lambda m, v: (self.sampled[slot][m][:k], None, None), ch.numpy())
            else:
                yield self._build(idx, vids, lambda m, v: (self.staged[slot][m][:k], None, self.staged_len[slot][m][:k]),
                                  ch.numpy())
