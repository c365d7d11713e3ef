"""One optimisation step as a replayable launch sequence (used by misc/run.py:run_train and bench.py).

The reference's step is five host statements (misc/run.py:254-261): zero_grad, forward, get_loss, backward,
clip_grad_value_ + optimizer.step.  On an MI355X that step is ~4 ms of device work behind ~350 kernel launches, so the
host must not be in the loop: `TrainStep` runs the first `eager_steps` steps launch by launch (they are real steps:
workspaces and the criterion's fused plan are created there), then captures the sequence ONCE into hipGraphs whose
inputs are static device buffers, and afterwards every step is: copy the batch into the static buffers, bump the
learning rate / host-side meters, replay.

Everything data dependent stays on the device inside the captured sequence: live-row lists (count in device memory),
dropout and masking seeds (Philox {seed, step} advanced by a kernel), Adam's step counter and learning rate
(`FusedAdam.step_dev / lr_dev`), the criterion's meters (`Criterion._meters`).

N > 1 ranks (runtime/ddp.py): the backward pass is split at the decoder's output and at the encoder outputs into
three graphs; each gradient bucket (vocabulary projection | decoder side | encoder side) is all-reduced (RCCL,
torch.distributed's own stream -- never captured) while the next backward graph replays, the last one after it; the decoder-side parameters are updated (their own Adam graph) while the second bucket is
still in flight, the encoder-side ones after it.  With opt['sync_bn'] the forward and the encoder-side backward reach
three more collectives (the BatchNorm statistics of the global batch): every stage is therefore captured as a SEQUENCE
of hipGraphs cut at those points (`_capture_stage`), and the replay issues the collectives between them.

A batch whose shapes differ from the captured ones (the ragged tail of an epoch) runs launch by launch.
"""
import os
import sys

import torch

from . import ops


BATCH_KEYS = ('feats', 'feats_a', 'feats_m', 'feats_i', 'tokens', 'tokens_1', 'labels', 'labels_1', 'category',
              'length_target')         # what misc/run.py:get_forword_results reads from a batch


def _tensors(batch, keys=BATCH_KEYS):
    """(key, index) -> tensor for the tensor / list-of-tensors entries of a batch dictionary the forward consumes"""
    out = {}
    for k in sorted(batch):
        if keys is not None and k not in keys:
            continue
        v = batch[k]
        if torch.is_tensor(v):
            out[(k, None)] = v
        elif isinstance(v, (list, tuple)) and len(v) and all(torch.is_tensor(x) for x in v):
            for i, x in enumerate(v):
                out[(k, i)] = x
    return out


def _signature(batch, keys=BATCH_KEYS):
    return tuple((k, tuple(t.shape), t.dtype) for k, t in _tensors(batch, keys).items())


class TrainStep(object):
    def __init__(self, model, crit, optimizer, forward, ddp=None, graph='auto', eager_steps=2, keys=BATCH_KEYS):
        """forward(batch) -> the results dictionary `crit.get_loss` consumes (labels included);
        optimizer: misc.optim.ScheduledOptim (or a bare FusedAdam); graph: 'auto' | 'on' | 'off';
        keys: the batch entries `forward` reads (they get static device buffers; everything else rides along)."""
        self.keys = keys
        assert graph in ('auto', 'on', 'off')
        self.model, self.crit, self.forward, self.ddp = model, crit, forward, ddp
        self.sched = optimizer if hasattr(optimizer, 'step_update_learning_rate') else None
        self.adam = optimizer._optimizer if self.sched is not None else optimizer
        self.graph_mode, self.eager_steps = graph, int(eager_steps)
        self.multi = ddp is not None and (ddp.world > 1 or ddp.force)
        self.split = ddp.bucket_split() if self.multi else None
        self.staged = self.split is not None
        # Two backward stages (decoder side incl. the vocabulary projection | encoder side) by default: each stage's
        # weight-gradient GEMMs run as ONE grouped launch at its end (ops.dw_group), and a group has to be large to need
        # few reduce splits -- a third stage for the vocabulary projection alone (NACF_DDP_STAGES=3: its bucket leaves
        # one stage earlier) costs more in the grouped launches than the earlier all-reduce start gains at one rank
        three_ok = self.staged and os.environ.get("NACF_DDP_STAGES", "2") == "3"
        self.hsplit = ddp.head_split() if three_ok else None
        self.three = None                   # decided at the first step (needs model._cut_head of a fused-loss forward)
        self.grad_scale = ddp.grad_scale if self.multi else 1.0
        # NACF_DDP_GRAPH_COLLECTIVES=1: capture the RCCL calls INSIDE the step graph (torch.distributed's NCCL/RCCL process
        # group is capturable: its internal stream forks from / joins the capturing stream through events), so that an
        # N > 1 step is ONE graph launch like the single-GPU step instead of 4-7 graphs with host-issued collectives
        # between them (+0.16 .. +0.27 ms per 2.8 ms step at one rank).  Off by default: it has only ever run with a
        # 1-rank group (no multi-GPU node here) -- flip it the moment one exists (tools/scale_check.sh).
        self.graph_collectives = self.multi and os.environ.get("NACF_DDP_GRAPH_COLLECTIVES", "0") == "1"
        # the Adam walk of step i leaves the gradient buffer zeroed for step i + 1 (one fill of 74 MB less per step); between
        # its steps the engine owns that buffer.  NACF_FUSED_ZERO_GRAD=0: fill at the start of every step as misc/run.py does
        self.fused_zero_grad = os.environ.get("NACF_FUSED_ZERO_GRAD", "1") != "0"
        self._grad_clean = False
        self._fwd_seen = None               # FlatParams.train_forwards after this engine's last step
        self._one = self._graph_loss = None
        self.static = self.sig = None
        self.loss = None                    # device scalar: the last step's loss
        self.n_steps = 0
        self.graphs = None                  # (front, decoder-side backward | None, encoder-side backward | None, adam)
        self.count_delta = None
        self._hold = {}

    # ---- the launch sequence -------------------------------------------------------------------------------------
    def _front(self, b):
        # the previous step's Adam walk left the gradients zeroed (FusedAdam.step(zero_grad=True)): no fill then
        if not self._grad_clean:
            self.adam.zero_grad()
        self._grad_clean = False
        loss = self.crit.get_loss(self.forward(b))
        if self.staged and self.three is None:
            self.three = self.hsplit is not None and bool(getattr(self.model, '_cut_head', None))
        # every backward stage queues the split-K combines of its weight-gradient GEMMs and runs them in one launch at
        # its end (ops.dw_group): a stage's gradient bucket is complete -- ready for its all-reduce -- when it returns
        with ops.dw_group():
            if self.staged and self.three:
                self._hold['hcut'], self._hold['hgrads'] = self.ddp.backward_head(loss)
            elif self.staged:
                self._hold['cut'], self._hold['grads'] = self.ddp.backward_to_cut(loss)
            else:
                loss.backward(self._one)    # (a resident 1.0: backward() would fill a fresh one every step)
        # a reference, not a copy: under capture this is the graph's own output buffer, alive as long as it is held here
        self.loss = loss.detach()

    def _mid(self):
        with ops.dw_group():
            self._hold['cut'], self._hold['grads'] = self.ddp.backward_mid(self._hold['hcut'], self._hold['hgrads'])

    def _back(self):
        with ops.dw_group():
            self.ddp.backward_from_cut(self._hold['cut'], self._hold['grads'])

    def _update(self, part=None):
        """part None: every parameter; 0: the late bucket flat[split:] (starts the step), 1: the rest flat[:split]"""
        z = self.fused_zero_grad
        if part is None:
            self.adam.step(grad_scale=self.grad_scale, zero_grad=z)
        elif part == 0:
            self.adam.step(grad_scale=self.grad_scale, lo=self.split, hi=None, bump=True, zero_grad=z)
        else:
            self.adam.step(grad_scale=self.grad_scale, lo=0, hi=self.split, bump=False, zero_grad=z)
        if part != 0:
            self._grad_clean = z          # every part of the flat gradient has been walked

    def _reduce_around(self, mid_stage, last_stage, update_late, update_early):
        """Gradient buckets leave as soon as they are complete and travel (RCCL's own stream, in this order) under the
        backward stages that follow: [vocabulary projection | under the decoder's backward,] decoder side | under the
        encoder's backward, encoder side | under the Adam update of everything that has already landed."""
        n = self.model.flat.grad.numel()
        works = []
        if self.three:
            works.append(self.ddp.all_reduce_range(self.hsplit, n))
            mid_stage()
            works.append(self.ddp.all_reduce_range(self.split, self.hsplit))
        else:
            works.append(self.ddp.all_reduce_range(self.split, n))
        last_stage()
        w_last = self.ddp.all_reduce_range(0, self.split)
        for w in works:
            if w is not None:
                w.wait()
        update_late()
        if w_last is not None:
            w_last.wait()
        update_early()

    def _eager(self, b):
        self._front(b)
        if self.staged:
            self._reduce_around(self._mid, self._back, lambda: self._update(0), lambda: self._update(1))
            self._hold.clear()
            return
        if self.multi:
            self.ddp.all_reduce_gradients()
        self._update()

    def _capture_stage(self, fn, pool, mode, breaks=True):
        """Run `fn` under stream capture.  The result is a launch SEQUENCE: hipGraphs cut wherever the code reached a
        collective inside forward / backward (DataParallel.all_reduce: the SyncBN statistics) -- RCCL calls are never
        captured, the replay issues them between the graphs.  All graphs share one memory pool (the autograd graph
        and its saved tensors live across the cuts)."""
        seq = []
        cur = [None]

        def begin():
            g = torch.cuda.CUDAGraph()
            if pool[0] is None:
                g.capture_begin(capture_error_mode=mode)
            else:
                g.capture_begin(pool=pool[0], capture_error_mode=mode)
            cur[0] = g

        def end():
            g = cur[0]
            g.capture_end()
            if pool[0] is None:
                pool[0] = g.pool()
            seq.append(g)
            cur[0] = None

        def brk(t):
            end()
            seq.append(t)
            begin()
        if self.ddp is not None and breaks:
            self.ddp._capture_break = brk
        begin()
        try:
            fn()
        finally:
            if cur[0] is not None:
                end()
            if self.ddp is not None:
                self.ddp._capture_break = None
        return seq

    def _run_seq(self, seq):
        for item in seq:
            if isinstance(item, torch.cuda.CUDAGraph):
                item.replay()
            elif isinstance(item, tuple):   # ('gather', out, t): an all-gather recorded at this point of the step
                self.ddp.all_gather(item[1], item[2])
            else:                           # a tensor: the all-reduce recorded at this point of the step
                self.ddp.all_reduce(item)

    def _own_forwards(self, count_before):
        """the forwards the capture itself ran are this engine's own: without this the first replay sees a "foreign" forward and
        fills the gradient buffer for nothing.  Only the capture's increments are taken over: a foreign forward
        that happened BEFORE the capture stays visible to the guard."""
        flat = getattr(self.model, 'flat', None)
        if flat is not None and self._fwd_seen is not None and count_before is not None:
            self._fwd_seen += flat.train_forwards - count_before

    def _capture(self):
        dev = self.loss.device
        before = list(self.crit._loss_cnt)
        fwd_before = getattr(getattr(self.model, 'flat', None), 'train_forwards', None)
        # drain first: RCCL's watchdog thread polls the events of unfinished collectives, which is illegal while a
        # capture is open in another thread ("thread_local" below keeps unrelated threads out of it as well)
        torch.cuda.synchronize(dev)
        # 'thread_local' keeps unrelated threads out of the capture.  With SyncBN the capture is cut and re-opened INSIDE
        # the backward pass, i.e. on autograd's device thread, and only a 'relaxed' capture may be ended by another
        # thread than the one that began it (hipErrorStreamCaptureWrongThread otherwise).
        jr = getattr(self.model, 'joint_representation_learner', None)
        mode = 'relaxed' if (self.multi and getattr(jr, '_sync', None) is not None) else 'thread_local'
        pool = [None]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        if self.graph_collectives:
            def whole():
                self._front(self.static)
                if self.staged:
                    self._reduce_around(self._mid, self._back, lambda: self._update(0), lambda: self._update(1))
                    self._hold.clear()
                else:
                    self.ddp.all_reduce_gradients()
                    self._update()
            with torch.cuda.stream(side):
                one = self._capture_stage(whole, pool, 'relaxed', breaks=False)
            torch.cuda.current_stream(dev).wait_stream(side)
            self.count_delta = [a - b for a, b in zip(self.crit._loss_cnt, before)]
            self.crit._loss_cnt = before
            self.graphs = (one, None, None, None)
            self._graph_images = self._image_state()
            self._graph_loss = self.loss
            self._own_forwards(fwd_before)
            return
        with torch.cuda.stream(side):
            def front_fn():
                self._front(self.static)
                if not self.multi:
                    self._update()
            front = self._capture_stage(front_fn, pool, mode)
            back = upd = mid = None
            if self.staged:                 # same memory pool: the autograd graph of `front` is still alive
                if self.three:
                    mid = self._capture_stage(self._mid, pool, mode)
                back = self._capture_stage(self._back, pool, mode)
                self._hold.clear()
                upd = [self._capture_stage(lambda part=part: self._update(part), pool, mode) for part in (0, 1)]
            elif self.multi:
                upd = self._capture_stage(self._update, pool, mode)
        torch.cuda.current_stream(dev).wait_stream(side)
        # capturing ran the host side of get_loss once without executing anything: take its sample-count increments
        # as the per-replay delta and undo them
        self.count_delta = [a - b for a, b in zip(self.crit._loss_cnt, before)]
        self.crit._loss_cnt = before
        self.graphs = (front, mid, back, upd)
        self._graph_images = self._image_state()
        self._graph_loss = self.loss
        self._own_forwards(fwd_before)

    def _image_state(self):
        """what the captured launches bake in besides the static buffers: the GEMM mode and the weight-image set"""
        flat = getattr(self.model, 'flat', None)
        return (ops.gemm_mode(), None if flat is None else flat.image_epoch)

    def _replay(self):
        front, mid, back, upd = self.graphs
        run = self._run_seq
        self.loss = self._graph_loss
        run(front)
        if self.graph_collectives:
            pass                    # the one graph holds the reductions and the update
        elif self.staged:
            self._reduce_around((lambda: run(mid)) if mid is not None else None, lambda: run(back),
                                lambda: run(upd[0]), lambda: run(upd[1]))
        else:
            if self.multi:
                self.ddp.all_reduce_gradients()
            if upd is not None:
                run(upd)
        self.crit._loss_cnt = [c + d for c, d in zip(self.crit._loss_cnt, self.count_delta)]
        # the replayed Adam walk wrote the fp32 master weights on the device; the graph refreshed the weight images BEFORE
        # it, so they are one step behind: a direct decoder / vocabulary / Translator call must rebuild them (FlatParams.ensure_images)
        flat = getattr(self.model, 'flat', None)
        if flat is not None:
            flat.touch()

    def _graph_ready(self):
        """the criterion must keep ALL its running sums in the in-place device meter vector (fused form)"""
        c = self.crit
        return getattr(c, '_meters', None) is not None and all(s is None for s in c._loss_sum) \
            and c._acc is None and c._ppl is None

    # ---- one step ------------------------------------------------------------------------------------------------
    def __call__(self, batch=None):
        """batch=None: step again on the contents of the static buffers (bench: one resident synthetic batch)"""
        if self.static is None:
            assert batch is not None
            self.static = dict(batch)
            for (k, i), t in _tensors(batch, self.keys).items():
                if i is None:
                    self.static[k] = t.clone()
                else:
                    self.static[k] = list(self.static[k])
                    self.static[k][i] = t.clone()
            # the two decoding passes of NACF / ARB2 are batched as 2B rows: keep their tokens (and labels) back to
            # back in ONE buffer so that the model takes a view instead of concatenating every step
            for first, second in (('tokens_1', 'tokens'), ('labels_1', 'labels')):
                a, b = self.static.get(first), self.static.get(second)
                if torch.is_tensor(a) and torch.is_tensor(b) and a.shape == b.shape and a.dtype == b.dtype:
                    pair = torch.stack([a, b])
                    self.static[first], self.static[second] = pair[0], pair[1]
            self.sig = _signature(batch, self.keys)
            self.loss = torch.zeros((), device=next(iter(_tensors(batch, self.keys).values())).device)
            self._one = torch.ones((), device=self.loss.device)
        self.n_steps += 1
        if self.sched is not None:
            self.sched.step_update_learning_rate()      # host counter + one fill of the device-side learning rate
        if batch is not None and _signature(batch, self.keys) != self.sig:
            self._foreign_backward_guard()
            self._eager(batch)                          # ragged tail of an epoch
            self._fwd_seen = getattr(getattr(self.model, 'flat', None), 'train_forwards', None)
            return None
        if batch is not None:
            dst = _tensors(self.static, self.keys)
            for key, src in _tensors(batch, self.keys).items():
                if dst[key].data_ptr() != src.data_ptr():
                    dst[key].copy_(src)
            for k, v in batch.items():                  # everything else (video ids, sample indices ...) rides along
                if (k, None) not in dst and (k, 0) not in dst:
                    self.static[k] = v
        if self.graphs is not None and self._image_state() != self._graph_images:
            # the process-wide GEMM mode changed, or the weight images were replaced (an eager forward in another mode):
            # the graphs would refresh and read the retired image set.  Drop them; this step runs launch by launch and
            # builds the new images, the next one re-captures.
            self.graphs = None
            self._hold.clear()
            self.n_steps = min(self.n_steps, self.eager_steps)
        if self.graphs is None and self.graph_mode != 'off' and self.n_steps > self.eager_steps:
            if self._graph_ready():
                try:
                    self._capture()
                except Exception as e:  # noqa: BLE001
                    if self.graph_mode == 'on':
                        raise
                    print('[nacf_amd] hipGraph capture failed (%s: %s); stepping launch by launch'
                          % (type(e).__name__, e), file=sys.stderr)
                    self.graphs, self.graph_mode = None, 'off'
                    self._hold.clear()
                    torch.cuda.synchronize()
            elif self.graph_mode == 'on':
                raise RuntimeError('nacf_amd: hipGraph capture needs the fused criterion (opt["fused_loss"])')
            else:
                self.graph_mode = 'off'
        self._foreign_backward_guard()
        if self.graphs is not None:
            self._replay()
        else:
            self._eager(self.static)
        self._fwd_seen = getattr(getattr(self.model, 'flat', None), 'train_forwards', None)

    def _foreign_backward_guard(self):
        """the fused zero-grad leaves the gradient buffer clean for THIS engine's next step, and a captured step
        has no fill in it.  A training forward that did not come from here (a manual loss.backward(), misc/run.py's
        launch-by-launch path, a second engine) may have summed into that buffer since: fill it, as optimizer.zero_grad() would.
        What is tracked: training forwards through Seq2Seq (flat.train_forwards); a backward reached through a direct call of a
        sub-module (model.decoder(...)) with grad enabled is not seen."""
        flat = getattr(self.model, 'flat', None)
        if flat is None or self._fwd_seen is None or flat.train_forwards == self._fwd_seen or not self._grad_clean:
            return
        self._grad_clean = False
        if self.graphs is not None:
            self.adam.zero_grad()
            self._grad_clean = True         # (what the replayed step assumes; its Adam walk re-establishes it)

    @property
    def captured(self):
        return self.graphs is not None
