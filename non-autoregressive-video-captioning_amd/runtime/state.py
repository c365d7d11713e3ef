"""Per-model runtime state: flat fp32 parameter / gradient buffers (one RCCL
bucket, one fused Adam launch), dropout RNG state, dropout-site salts."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from .functional import Pack
from .ops import RngState


class Runtime:
    def __init__(self, seed: int = 0):
        self.seed = int(seed)
        self._rng: Optional[RngState] = None
        self._salt = 0

    def next_salt(self) -> int:
        self._salt += 1
        return (self._salt * 0x9E3779B1 + 0x7F4A7C15) & 0xFFFFFFFF

    def rng(self, device) -> RngState:
        if self._rng is None or self._rng.state.device != torch.device(device):
            self._rng = RngState(self.seed, device)
        return self._rng

    def advance(self) -> None:
        if self._rng is not None:
            self._rng.advance()


class FlatParams:
    """Re-homes every parameter of a model into one flat fp32 buffer (and its
    gradient into a second one).  ``groups`` lists parameters that must be
    adjacent, in order, so packed views (w1|w2, q|k|v, k|v) are plain slices."""

    ALIGN = 8  # floats (32 B)

    def __init__(self, groups: Sequence[Sequence[nn.Parameter]]):
        seen = set()
        layout: List[List[nn.Parameter]] = []
        for g in groups:
            g = [p for p in g if id(p) not in seen]
            for p in g:
                seen.add(id(p))
            if g:
                layout.append(g)
        device = layout[0][0].device
        self.offset: Dict[int, int] = {}
        off = 0
        for g in layout:
            off = (off + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            for p in g:
                assert p.dtype == torch.float32, "fp32 parameters only"
                self.offset[id(p)] = off
                off += p.numel()
        self.total = (off + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.data = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.params: List[nn.Parameter] = [p for g in layout for p in g]
        self.image_specs: List[tuple] = []      # (offset, N, K, want_transposed) of every GEMM weight (pack(..., image=))
        self.images = None                      # ops.WeightImages of the bf16 GEMM modes
        # captured hipGraphs bake the images' addresses in (the refresh launch and every GEMM's P operand): replaced image
        # sets are RETIRED (unregistered, memory kept) rather than freed, and every replacement bumps `image_epoch`, which
        # the graph owners (runtime/engine.py:TrainStep, decoding/na_generate.py) compare before a replay
        self.image_epoch = 0
        self._retired_images: List[object] = []
        # `version` counts writes to the fp32 master weights that went through this package (optimiser step,
        # load_state_dict, broadcast); `images_version` is the version the images were last built from
        self.version = 0
        # training-mode forward passes through Seq2Seq (whoever called them): the step engine owns the gradient buffer
        # between ITS steps only as long as nobody else ran one (runtime/engine.py: the fused zero-grad)
        self.train_forwards = 0
        self.images_version = -1
        with torch.no_grad():
            for p in self.params:
                o, n = self.offset[id(p)], p.numel()
                v = self.data[o:o + n].view(p.shape)
                v.copy_(p.data)
                p.data = v
        self.attach_grads(zero=False)

    def attach_grads(self, zero: bool) -> None:
        if zero:
            self.grad.zero_()
        for p in self.params:
            o, n = self.offset[id(p)], p.numel()
            if p.requires_grad:
                p.grad = self.grad[o:o + n].view(p.shape)

    def grads_attached(self) -> bool:
        for p in self.params:
            if p.requires_grad:
                o = self.offset[id(p)]
                return p.grad is not None and p.grad.data_ptr() == self.grad[o:].data_ptr()
        return True

    def sync_images(self) -> None:
        """bf16 GEMM modes: (re)build the weight images from the current fp32 values.  Called at every forward entry
        (models/seq2seq.py), so the GEMMs of that forward AND of its backward read images of exactly the weights the
        fp32 kernels would read; one ~40 us launch (captured with the step).  Nothing to do in the fp32 mode."""
        from . import ops
        mode = ops.gemm_mode()
        if mode == 0 or not self.data.is_cuda:
            if self.images is not None:
                self._retire_images()
            return
        if self.images is None or self.images.ns != mode or self.images.flat.data_ptr() != self.data.data_ptr():
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("nacf_amd: weight images must exist before a hipGraph capture (run a warm-up step)")
            if self.images is not None:
                self._retire_images()
            self.images = ops.WeightImages(self.data, self.image_specs, mode)
            self.image_epoch += 1
        self.images.refresh()
        self.images_version = self.version
        self._images_tensor_version = self.data._version

    def _retire_images(self) -> None:
        self.images.close()                          # unregister: no new launch finds them
        self._retired_images.append(self.images)     # ... but a graph captured earlier may still read / refresh them
        self.images = None
        self.image_epoch += 1

    def touch(self) -> None:
        """the fp32 master weights were written (optimiser step, load_state_dict, parameter broadcast)"""
        self.version += 1

    def ensure_images(self) -> None:
        """entry points that run GEMMs WITHOUT going through Seq2Seq.encode (a direct model.decoder(...) /
        vocab_logprobs / Translator call on cached encoder outputs): rebuild the images if the weights changed since
        they were built (FlatParams.version, or the flat tensor's autograd version for writes made with torch ops)."""
        from . import ops
        if not self.data.is_cuda or torch.cuda.is_current_stream_capturing():
            return
        # writes that bypass this package (torch.optim on model.parameters(), p.data.copy_, nn.init, an EMA swap, a sub-module
        # load_state_dict) are seen through the flat tensor's autograd version counter, which every parameter view shares:
        # an O(1) host check.  The library's own kernels write through raw pointers: those call touch().
        tv = self.data._version
        stale = (self.images is None or self.images.ns != ops.gemm_mode() or self.images_version != self.version
                 or getattr(self, '_images_tensor_version', None) != tv)
        if stale if ops.gemm_mode() != 0 else self.images is not None:      # (fp32 mode: an existing image set is retired, as at a forward entry)
            self.sync_images()

    def pack(self, ws: Sequence[nn.Parameter], bs: Optional[Sequence[nn.Parameter]] = None,
             image: Optional[str] = None) -> Pack:
        """Packed (row-concatenated) view of adjacent weights [+ biases].  image: 'fwd' (the matrix is the P operand
        of a forward GEMM) or 'both' (also of a dX GEMM): listed for the bf16 weight images."""
        if image is not None and ws and ws[0] is not None and ws[0].dim() == 2:
            spec = (self.offset[id(ws[0])], sum(p.shape[0] for p in ws), ws[0].shape[1], image == 'both')
            if spec[:3] not in [s_[:3] for s_ in self.image_specs]:
                self.image_specs.append(spec)
        def cat(ps):
            if ps is None or len(ps) == 0 or ps[0] is None:
                return None, None
            o0 = self.offset[id(ps[0])]
            o = o0
            for p in ps:
                assert self.offset[id(p)] == o, "parameters of a pack must be adjacent in the flat buffer"
                o += p.numel()
            rows = sum(p.shape[0] for p in ps)
            shape = (rows,) + tuple(ps[0].shape[1:])
            gv = self.grad[o0:o].view(shape) if all(p.requires_grad for p in ps) else None
            return self.data[o0:o].view(shape), gv
        w, gw = cat(ws)
        b, gb = cat(bs)
        return Pack(w, b, gw, gb)
