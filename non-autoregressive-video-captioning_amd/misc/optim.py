"""Optimiser for the training step (reference: misc/optim.py:3-68 plus the
`clip_grad_value_` of misc/run.py:260): elementwise clip to +-grad_clip, Adam
with L2-in-gradient weight decay on every parameter, lr * min(step/(warmup+1),
1), and a per-epoch x`decay` floor-ed at `minimum_learning_rate`.

One `nacf_adam_step` launch walks the model's flat parameter / gradient / moment
buffers (~0.5 GB of HBM traffic per step at 18.5 M parameters); the step count
and the learning rate live in device memory so a captured hipGraph can replay it.
"""
import torch

from ..runtime import ops


class FusedAdam(object):
    def __init__(self, model, weight_decay=5e-4, betas=(0.9, 0.999), eps=1e-8, grad_clip=5.0):
        self.model = model
        self.wd, self.betas, self.eps, self.grad_clip = weight_decay, betas, eps, grad_clip
        self._alloc()

    def _alloc(self):
        flat = self.model.flat
        self._flat = flat
        self.exp_avg = torch.zeros_like(flat.data)
        self.exp_avg_sq = torch.zeros_like(flat.data)
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=flat.data.device)
        self.lr_dev = torch.zeros(1, dtype=torch.float32, device=flat.data.device)
        self._lr_on_dev = 0.0
        self.param_groups = [{'lr': 0.0, 'params': flat.params}]

    def set_lr(self, lr):
        self.param_groups[0]['lr'] = lr
        if lr != self._lr_on_dev:            # (constant after the warm-up: no fill launch per step then)
            self.lr_dev.fill_(lr)
            self._lr_on_dev = lr

    def zero_grad(self):
        self.model.zero_grad()

    def step(self, grad_scale=1.0, lo=None, hi=None, bump=True, zero_grad=False):
        """lo / hi: update only flat[lo:hi] (one gradient bucket of the data-parallel step); exactly one part of a
        step passes bump=True, and it must come first.  zero_grad: leave the gradients of the updated range zeroed
        (runtime/engine.py: the NEXT step's optimizer.zero_grad() for 4 more bytes per parameter of this walk; the
        reference's optimizer.step() leaves them, so this is off unless the step engine asks)"""
        self.model.flat.touch()      # (weight images are rebuilt at the next forward entry)
        if self._flat is not self.model.flat:   # model moved (.to/.cuda) after the optimiser was built
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('nacf_amd: the model was re-homed during a hipGraph capture')
            old = (self.exp_avg, self.exp_avg_sq, self.step_dev)
            self._alloc()
            if old[0].numel() == self.exp_avg.numel():      # same layout on a new device: the Adam state moves along
                self.exp_avg.copy_(old[0])
                self.exp_avg_sq.copy_(old[1])
                self.step_dev.copy_(old[2])
            self.lr_dev.fill_(self.param_groups[0]['lr'])
            self._lr_on_dev = self.param_groups[0]['lr']
        f = self._flat
        sl = slice(lo, hi)
        ops.adam_step(f.data[sl], f.grad[sl], self.exp_avg[sl], self.exp_avg_sq[sl], self.lr_dev, self.step_dev,
                      self.betas[0], self.betas[1], self.eps, self.wd, self.grad_clip, grad_scale, bump=bump, zero_grad=zero_grad)


class FusedRMSprop(FusedAdam):
    """torch.optim.RMSprop as the reference constructs it (misc/optim.py:52-60: defaults alpha 0.99, eps 1e-8, momentum 0, not
    centred, weight_decay from the options) on the flat buffers: one nacf_rmsprop_step walk per bucket, the same clip / gradient
    scale / fused zero-grad contract as FusedAdam (the step engine drives both through .step())."""

    def __init__(self, model, weight_decay=5e-4, alpha=0.99, eps=1e-8, grad_clip=5.0):
        self.alpha = alpha
        super().__init__(model, weight_decay=weight_decay, eps=eps, grad_clip=grad_clip)

    def _alloc(self):
        super()._alloc()
        self.exp_avg = None              # (no first moment: momentum 0)

    def step(self, grad_scale=1.0, lo=None, hi=None, bump=True, zero_grad=False):
        self.model.flat.touch()
        if self._flat is not self.model.flat:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('nacf_amd: the model was re-homed during a hipGraph capture')
            old = self.exp_avg_sq
            self._alloc()
            if old.numel() == self.exp_avg_sq.numel():
                self.exp_avg_sq.copy_(old)
            self.lr_dev.fill_(self.param_groups[0]['lr'])
            self._lr_on_dev = self.param_groups[0]['lr']
        f = self._flat
        sl = slice(lo, hi)
        ops.rmsprop_step(f.data[sl], f.grad[sl], self.exp_avg_sq[sl], self.lr_dev, self.alpha, self.eps, self.wd, self.grad_clip,
                         grad_scale, zero_grad=zero_grad)


class ScheduledOptim(object):
    def __init__(self, optimizer, learning_rate, minimum_learning_rate, epoch_decay_rate, grad_clip=2,
                 n_warmup_steps=0, summarywriter=None):
        self._optimizer = optimizer
        self.n_current_steps = 0
        self.lr, self.mlr, self.decay = learning_rate, minimum_learning_rate, epoch_decay_rate
        self.grad_clip = grad_clip
        self.n_warmup_steps = n_warmup_steps

    def step(self, grad_scale=1.0):
        self.step_update_learning_rate()
        self._optimizer.step(grad_scale=grad_scale)

    def zero_grad(self):
        self._optimizer.zero_grad()

    def epoch_update_learning_rate(self):
        if self.n_current_steps > self.n_warmup_steps:
            self.lr = max(self.mlr, self.decay * self.lr)

    def step_update_learning_rate(self):
        self.n_current_steps += 1
        ratio = min(self.n_current_steps / (self.n_warmup_steps + 1.0), 1)
        self._optimizer.set_lr(self.lr * ratio)

    def get_lr(self):
        return self.lr


def get_optimizer(opt, model, summarywriter=None):
    kind = opt['optim'].lower()
    assert kind in ('adam', 'rmsprop'), kind          # misc/optim.py:52-57: the reference's two optimisers
    cls = FusedAdam if kind == 'adam' else FusedRMSprop
    inner = cls(model, weight_decay=opt['weight_decay'], grad_clip=opt.get('grad_clip', 5.0))
    return ScheduledOptim(inner, learning_rate=opt['learning_rate'],
                          minimum_learning_rate=opt['minimum_learning_rate'], epoch_decay_rate=opt['decay'],
                          grad_clip=opt.get('grad_clip', 5.0), n_warmup_steps=opt.get('n_warmup_steps', 0))
