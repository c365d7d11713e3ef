"""Data-parallel engine: one process per GPU, RCCL over xGMI through
torch.distributed (backend "nccl" IS RCCL on ROCm; "gloo" for the CPU tests).

The reference has no distributed code at all; semantics here
are "same as one process with the global batch": every rank holds a full
replica, takes a contiguous shard of the global batch, and after backward the
flat fp32 gradient buffer (73.8 MB for NACF/MSRVTT-shape; two or three buckets, see "Overlap" below) is
all-reduced (sum); the 1/world scale is folded into the fused Adam launch, so
clip(+-5) follows the reduce exactly as misc/run.py:258-261 orders them.
Per-rank losses are normalised by the LOCAL batch (misc/crit.py:40), so the
mean of rank gradients equals the global-batch gradient.  BatchNorm (the one place where samples interact,
models/joint_representation.py:43-45): per-rank statistics by default; with opt['sync_bn'] the statistics are those of
the global batch -- the ranks exchange one [n_modalities, 512] vector per statistics pass (`sync_point`, three tiny
all-reduces per step) and N-rank training IS one process with the global batch (tests/test_ddp_cpu.py, world 2 and 4).

Overlap: the flat buffer is laid out encoder | fusion | length head | decoder | vocabulary projection, and backward
finishes the decoder side first.  `backward_to_cut` / `backward_from_cut` split the backward pass at the encoder
outputs; the all-reduce of the decoder-side bucket (61 of 74 MB) is launched between the two and runs on
torch.distributed's own stream while the encoder's backward (~1 ms of GEMMs) executes.  xGMI is point-to-point, so
at 2 GPUs a 74 MB ring step crosses ONE link: hiding it matters most at small N.
With the fused vocabulary loss there is a third bucket: `backward_head` stops at the decoder's output
(model._cut_head), at which point the vocabulary projection's gradients (the tail of the flat buffer, 21.6 MB) are
complete and leave under the decoder's backward (`backward_mid`); runtime/engine.py drives the sequence.
"""
import torch
import torch.distributed as dist


def shard_range(global_batch: int, rank: int, world: int):
    """contiguous shard [lo, hi) of the global batch owned by `rank`"""
    assert global_batch % world == 0, 'global batch must divide evenly over ranks'
    per = global_batch // world
    return rank * per, (rank + 1) * per


_host_broadcast_seq = [0]


def host_broadcast_int(value, tag, src=0, timeout_s=6 * 3600):
    """`value` of rank `src` on every rank, through the rendezvous store -- a HOST-side wait.  For the long one-sided phases
    of training (rank 0 decodes and scores the validation split, misc/run.py): a device collective posted by the idle
    ranks would sit in RCCL's queue for the whole evaluation, in reach of its watchdog timeout."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return int(value)
    from datetime import timedelta
    store = dist.distributed_c10d._get_default_store()
    # every rank calls this the same number of times in the same order, so a per-process call counter makes the key unique
    # per CALL: a second train_network_all on the same process group (teacher then student, a resume, a test calling it
    # twice) must not find the previous run's verdict under the same tag.  Keys are never deleted (a slow rank may not have
    # read one yet; they are a few bytes per evaluated epoch)
    seq = _host_broadcast_seq[0]
    _host_broadcast_seq[0] += 1
    key = 'nacf_amd/host_broadcast/%d/%s' % (seq, tag)
    if dist.get_rank() == src:
        store.set(key, str(int(value)))
        return int(value)
    store.wait([key], timedelta(seconds=timeout_s))
    return int(store.get(key))


class DataParallel(object):
    def __init__(self, model, process_group=None, force_collectives=False):
        self.model = model
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        # issue the collectives even for a 1-rank group (exercises the N>1 launch sequence on one GPU)
        self.force = bool(force_collectives) and dist.is_initialized()
        self._capture_break = None      # set by runtime/engine.py while it captures: ends a hipGraph at a collective
        self.n_sync_points = 0
        jr = getattr(model, 'joint_representation_learner', None)
        if jr is not None and getattr(jr, 'sync_bn', False):
            jr._sync = self if (self.world > 1 or self.force) else None

    # ---- collectives INSIDE forward / backward (SyncBN statistics)
    def all_reduce(self, t):
        """sum `t` over the ranks, ordered with the current stream.  While the engine captures a step this ends the
        running hipGraph, records the collective and opens the next graph (RCCL calls are never captured)."""
        self.n_sync_points += 1
        if self._capture_break is not None:
            self._capture_break(t)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def all_gather(self, out, t):
        """out[r] = rank r's `t` (out: [world, *t.shape]), ordered with the current stream; cuts a capture like all_reduce"""
        self.n_sync_points += 1
        if self._capture_break is not None:
            self._capture_break(('gather', out, t))
        elif t.is_cuda:
            dist.all_gather_into_tensor(out, t, group=self.group)
        else:                       # gloo (CPU tests)
            dist.all_gather(list(out.unbind(0)), t, group=self.group)

    def ragged_flag(self, device):
        """device int32[1], sticky: set by nacf_bn_sync_merge when a rank's row count (it travels inside the statistics' own
        all-gather) differs from this rank's -- the merged statistics of that step are NaN on every rank; read at the sampled
        host checks below and by raise_if_ragged()"""
        f = getattr(self, '_ragged_flag', None)
        if f is None or f.device != device:
            f = self._ragged_flag = torch.zeros(1, dtype=torch.int32, device=device)
        return f

    def raise_if_ragged(self):
        """host read of the sticky flag (a sync point: call it where the host reads the device anyway, e.g. with the meters)"""
        f = getattr(self, '_ragged_flag', None)
        if f is not None and int(f.item()) != 0:
            raise RuntimeError('nacf_amd: SyncBN saw ranks with different row counts (a ragged global batch): the statistics of '
                               'that step were NaN on every rank; shard the global batch evenly (runtime/ddp.py:shard_range)')

    def assert_equal_rows(self, n_rows, what='SyncBN'):
        """SyncBN forms the global statistics as if every rank held `n_rows` rows (runtime/functional.py:BNConcatFn: n_tot =
        rows x world; the reference's single process has one batch): a ragged global batch must fail loudly, not normalise
        with the wrong count.  EVERY step is guarded on the device: the row counts travel inside the statistics' all-gather and
        nacf_bn_sync_merge turns the merged statistics into NaN when they differ (no sampling of the contract).
        This host-side check only adds the descriptive error: a blocking all-reduce plus a host read, on the first four
        launch-by-launch calls and on every 64th after that -- a schedule that depends only on the CALL COUNT, which is the same
        on every rank, so ranks that disagree about the rows still all reach the collective -- and it reads the sticky device
        flag then.  Never inside a stream capture (a captured step has the shapes of the eager steps before it)."""
        if self.world == 1 or not dist.is_initialized():
            return
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            return
        self._rows_calls = getattr(self, '_rows_calls', 0) + 1
        if self._rows_calls > 4 and self._rows_calls % 64 != 0:
            return
        self.raise_if_ragged()
        dev = self.model.flat.data.device if hasattr(self.model, 'flat') else torch.device('cpu')
        t = torch.tensor([n_rows, -n_rows], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        hi, lo = int(t[0]), -int(t[1])
        if hi != lo:
            raise RuntimeError('nacf_amd: %s needs the same number of rows on every rank (this rank %d, ranks hold %d..%d): '
                               'shard the global batch evenly (runtime/ddp.py:shard_range)' % (what, n_rows, lo, hi))

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def broadcast_parameters(self, src=0):
        if self.world == 1 and not self.force:
            return
        dist.broadcast(self.model.flat.data, src, group=self.group)
        self.model.flat.touch()
        for b in self.model.buffers():
            dist.broadcast(b, src, group=self.group)

    def all_reduce_gradients(self):
        """sum-all-reduce of the single flat gradient bucket"""
        if self.world == 1 and not self.force:
            return
        dist.all_reduce(self.model.flat.grad, op=dist.ReduceOp.SUM, group=self.group)

    # ---- overlapped variant: two buckets, the big one travels while the encoder's backward still runs ----
    def bucket_split(self):
        """Offset `s` such that flat.grad[s:] holds exactly the gradients of model.late_parameters() (decoder +
        vocabulary projection, ~80 % of the bytes) and flat.grad[:s] the rest (encoder, fusion, length head);
        None when the flat layout does not separate them."""
        flat = self.model.flat
        late = {id(p) for p in self.model.late_parameters()}
        lo = min(flat.offset[i] for i in late)
        for p in flat.params:
            if (flat.offset[id(p)] >= lo) != (id(p) in late):
                return None
        return lo

    def head_split(self):
        """Offset `h` (> bucket_split()) such that flat.grad[h:] holds exactly the gradients of model.head_parameters()
        (the vocabulary projection: complete as soon as backward has passed the decoder's output); None if there is no
        such tail."""
        head_fn = getattr(self.model, 'head_parameters', None)
        s = self.bucket_split()
        if head_fn is None or s is None:
            return None
        head = {id(p) for p in head_fn()}
        if not head:
            return None
        flat = self.model.flat
        lo = min(flat.offset[i] for i in head)
        for p in flat.params:
            if (flat.offset[id(p)] >= lo) != (id(p) in head):
                return None
        return lo if lo > s else None

    def backward_head(self, loss):
        """Stage 0 of the three-stage backward: from the loss to the decoder's output / the length head's output
        (model._cut_head); writes the vocabulary-projection gradients."""
        cut = list(self.model._cut_head)
        head = [p for p in self.model.head_parameters() if p.requires_grad]
        grads = torch.autograd.grad(loss, cut + head, allow_unused=True)
        return cut, list(grads[:len(cut)])

    def backward_mid(self, head_cut, head_grads):
        """Stage 1 of three: from the decoder's output down to the encoder outputs (decoder + length head)."""
        cut = list(self.model._cut)
        head_ids = {id(p) for p in self.model.head_parameters()}
        late = [p for p in self.model.late_parameters() if p.requires_grad and id(p) not in head_ids]
        pairs = [(t, g) for t, g in zip(head_cut, head_grads) if g is not None]
        grads = torch.autograd.grad([t for t, _ in pairs], cut + late, grad_outputs=[g for _, g in pairs], allow_unused=True)
        return cut, list(grads[:len(cut)])

    def all_reduce_range(self, lo, hi, async_op=True):
        if self.world == 1 and not self.force:
            return None
        return dist.all_reduce(self.model.flat.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def backward_to_cut(self, loss):
        """Stage 1: backward from the loss down to the encoder outputs (model._cut).  The decoder-side parameters are
        listed as inputs too so that every Function that only leads to parameters (e.g. an embedding lookup) still
        runs -- the kernels write parameter gradients as a side effect and return None for them."""
        cut = list(self.model._cut)
        late = [p for p in self.model.late_parameters() if p.requires_grad]
        grads = torch.autograd.grad(loss, cut + late, allow_unused=True)
        return cut, list(grads[:len(cut)])

    @staticmethod
    def backward_from_cut(cut, grads):
        """Stage 2: the encoder side."""
        pairs = [(t, g) for t, g in zip(cut, grads) if g is not None]
        torch.autograd.backward([t for t, _ in pairs], [g for _, g in pairs])

    def all_reduce_bucket(self, which, async_op=True):
        """which = 0: flat.grad[split:] (complete after stage 1), 1: flat.grad[:split].  Returns the Work handle
        (None for a single process): torch.distributed runs it on its own stream, ordered after the work already
        queued on the current stream, so the caller can keep launching stage 2."""
        if self.world == 1 and not self.force:
            return None
        s = self.bucket_split()
        g = self.model.flat.grad
        view = g[s:] if which == 0 else g[:s]
        return dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
