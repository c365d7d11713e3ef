"""Tensor-level wrappers over the C ABI (no autograd here).

PyTorch is plumbing only: it owns device memory and the HIP stream; every
computation below is a libnacf_hip kernel launched on torch's current stream.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from . import lib as L

Tensor = torch.Tensor


def _stream() -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _chk_f32(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise L.NacfLibraryError("nacf ops need tensors on the HIP device (no CPU fallback)")
        if t.dtype != torch.float32:
            raise TypeError(f"expected float32, got {t.dtype}")


def _rows2d(t: Tensor):
    """(rows, cols, ld) of a 2-D view whose last dim is contiguous."""
    assert t.dim() == 2 and t.stride(1) == 1, (t.shape, t.stride())
    return t.shape[0], t.shape[1], t.stride(0)


# ---------------------------------------------------------------- workspace
class _Workspace:
    """One grow-only scratch buffer per device; kernels on one stream run in order, so consecutive ops can share it.

    hipGraph safety: a captured graph (the training step, a decode batch) bakes this buffer's address into its
    split-K / BatchNorm / soft-max-statistics launches.  Growing the buffer later (a bigger decode batch, a teacher
    model, a wider vocabulary) must therefore never free the old one: once any capture has seen a buffer it is
    RETIRED instead of released when a larger one replaces it -- replays keep writing into memory that still
    belongs to them, new launches use the new buffer.  Growth during a capture itself is an error (warm up first)."""

    def __init__(self):
        self.buf = {}
        self.in_graph = set()      # device keys whose current buffer has been handed out during a capture
        self.retired = []          # buffers captured graphs may still reference: kept alive for the process lifetime

    def get(self, nbytes: int, device) -> Tensor:
        key = device.index if device.index is not None else torch.cuda.current_device()
        b = self.buf.get(key)
        capturing = torch.cuda.is_current_stream_capturing()
        if b is None or b.numel() < nbytes:
            if capturing:
                raise L.NacfLibraryError("workspace growth during graph capture: run a warm-up step first")
            if b is not None and key in self.in_graph:
                self.retired.append(b)
                self.in_graph.discard(key)
            b = torch.empty(max(int(nbytes * 1.25), 1 << 20), dtype=torch.uint8, device=device)
            self.buf[key] = b
        if capturing:
            self.in_graph.add(key)
        return b

    def reserve(self, nbytes: int, device) -> None:
        """grow to at least nbytes now (before a capture), e.g. to the larger of the training and decoding needs"""
        self.get(int(nbytes), torch.device(device))


WORKSPACE = _Workspace()
# a second buffer for work that runs on a SIDE stream next to the main one (the length head, models/seq2seq.py): two
# streams must not share split-K scratch.  Selected by `with ops.aux_workspace():` around the launches (forward and
# backward bodies of the side-stream Function: autograd runs a device's backward nodes one after the other in one thread)
WORKSPACE_AUX = _Workspace()
_AUX_DEPTH = 0


class aux_workspace:
    def __enter__(self):
        global _AUX_DEPTH
        _AUX_DEPTH += 1
        return self

    def __exit__(self, *a):
        global _AUX_DEPTH
        _AUX_DEPTH -= 1
        return False


def _ws() -> _Workspace:
    return WORKSPACE_AUX if _AUX_DEPTH > 0 else WORKSPACE


class GemmProfiler:
    """HIP-event timing of every GEMM launch, used by bench.py for the live
    roofline figure.  The kernels run on torch's current stream, which is also
    where torch.cuda.Event records, so the pair brackets exactly the launch."""
    _LAYOUT = {0: ("true", "true"), 1: ("true", "false"), 2: ("false", "false")}

    def __init__(self):
        self.enabled = False
        self.records = []
        # grouped launches (the weight-gradient group, the encoder streams' wide group) as the REAL step runs them: with
        # `group_enabled` the groups stay on and every grouped launch is bracketed by events (`enabled` must be off: it
        # turns the grouping off to time one launch per call)
        self.group_enabled = False
        self.group_records = []      # (kernel name, [(M, N, K, rows)], event a, event b)

    def group_span(self, problems, launch):
        """run `launch()` (one grouped launch of `problems`), timed when group_enabled"""
        if not self.group_enabled:
            return launch()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = launch()
        b.record()
        name = (L.load().nacf_gemm_last_kernel() or b"?").decode()
        self.group_records.append((name, list(problems), a, b))
        return out

    def group_summary(self):
        out = {}
        for name, problems, a, b in self.group_records:
            r = out.setdefault(name, dict(calls=0, flops=0.0, ms=0.0, problems=0, single=True))
            r["calls"] += 1
            r["ms"] += a.elapsed_time(b)
            r["problems"] += len(problems)
            for M, N, K, rows in problems:
                m_live = min(M, int(rows.count)) if rows is not None else M
                r["flops"] += 2.0 * m_live * N * K
        return out

    def _splits(self, kind, M, N, K, flags=0):
        tile, splits = ctypes.c_int(0), ctypes.c_int(0)
        L.check(L.load().nacf_gemm_config(kind | flags, M, N, K, ctypes.byref(tile), ctypes.byref(splits)),
                "nacf_gemm_config")
        return splits.value

    def begin(self, kind, M, N, K, epi, rows=None, heavy=False):
        if not self.enabled:
            return None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        # kind 2 (dW: GEMM + split-K combine + bias column-sum) and EpiArgmax (GEMM + merge) spans hold
        # more than one kernel; only single-kernel spans are comparable with rocprof's per-kernel average
        splits = self._splits(kind, M, N, K, (0x100 if rows is not None else 0) | (0x200 if heavy else 0))
        single = kind != 2 and epi != "EpiArgmax" and splits == 1
        return [None, (M, N, K), a, b, single, rows, kind]

    def end(self, tok):
        if tok is not None:
            tok[3].record()
            # the template name of the kernel the entry point just launched (= rocprofv3's kernel name)
            tok[0] = (L.load().nacf_gemm_last_kernel() or b"?").decode()
            self.records.append(tok)

    def summary(self):
        out = {}
        for name, shape, a, b, single, rows, kind in self.records:
            r = out.setdefault(name, dict(calls=0, flops=0.0, dense_flops=0.0, ms=0.0, shapes=set(), single=single,
                                          by_shape={}))
            M, N, K = shape
            m_live = min(M, int(rows.count)) if rows is not None else M   # flops actually executed (live rows only)
            r["calls"] += 1
            r["flops"] += 2.0 * m_live * N * K
            r["dense_flops"] += 2.0 * M * N * K
            ms = a.elapsed_time(b)
            r["ms"] += ms
            r["single"] = r["single"] and single
            r["shapes"].add(shape)
            q = r["by_shape"].setdefault((kind,) + tuple(shape), dict(calls=0, flops=0.0, ms=0.0))
            q["calls"] += 1
            q["flops"] += 2.0 * m_live * N * K
            q["ms"] += ms
        return out


PROFILER = GemmProfiler()


# ---------------------------------------------------------------- GEMM arithmetic mode / weight images
def gemm_mode() -> int:
    """effective mode of the GEMM entry points: L.GEMM_F32 | L.GEMM_BF16 | L.GEMM_BF16X3 (env NACF_GEMM_MODE overrides)"""
    return int(L.load().nacf_gemm_get_mode())


def set_gemm_mode(mode) -> None:
    """process-wide: 'f32' (exact fp32 MFMA), 'bf16x3' (exact three-term split on the bf16 matrix cores), 'bf16'"""
    if isinstance(mode, str):
        if mode not in L.GEMM_MODE_BY_NAME:
            raise ValueError("gemm_mode must be one of %s (got %r)" % (sorted(L.GEMM_MODE_BY_NAME), mode))
        mode = L.GEMM_MODE_BY_NAME[mode]
    L.check(L.load().nacf_gemm_set_mode(int(mode)), "nacf_gemm_set_mode")


_FIXED_REG_SELFTEST = {"done": False}


def selftest_fixed_register_kernels(device) -> None:
    """the wide GEMM kernels keep their accumulators in registers the compiler is not told about;
    the build checks their disassembly (csrc/Makefile: wide-check), but a library built elsewhere, with another hipcc or
    other flags, would corrupt results silently.  Once per process, before the first exact-mode weight images of a model are
    built: one 512 x 512 x 256 product on the wide kernel (both wave-tile heights) against the 128 x 128 kernel, which keeps
    its accumulators in compiler-visible registers.  A mismatch beyond summation-order noise raises."""
    if _FIXED_REG_SELFTEST["done"] or os.environ.get("NACF_SELFTEST", "1") == "0":
        return
    if torch.cuda.is_current_stream_capturing() or PROFILER.enabled or wide_group.current is not None:
        return                                             # (not now: the next set of images tries again)
    _FIXED_REG_SELFTEST["done"] = True
    saved = {k: os.environ.get(k) for k in ("NACF_GEMM_TILE", "NACF_GEMM_WIDE", "NACF_GEMM_MODE")}
    for k in saved:
        os.environ.pop(k, None)
    mode = gemm_mode()                                     # the PROCESS mode (the environment override is out of the way)
    try:
        g = torch.Generator().manual_seed(1234)
        M, N, K = 512, 512, 256
        x = (torch.rand(M, K, generator=g) - 0.5).to(device)
        flat = ((torch.rand(N * K, generator=g) - 0.5) * 0.2).to(device)
        w = flat.view(N, K)
        set_gemm_mode("bf16x3")
        imgs = WeightImages(flat, [(0, N, K, False)], 3, _selftest=True)
        imgs.refresh()
        outs = {}
        for tag, env in (("tile128", {"NACF_GEMM_TILE": "128"}), ("wide1", {"NACF_GEMM_WIDE": "1"}), ("wide2", {"NACF_GEMM_WIDE": "2"})):
            for k in ("NACF_GEMM_TILE", "NACF_GEMM_WIDE"):
                os.environ.pop(k, None)
            os.environ.update(env)
            y = torch.empty(M, N, device=device)
            linear_fwd(x, w, y, None)
            outs[tag] = (y, (L.load().nacf_gemm_last_kernel() or b"").decode())
        imgs.close()
        ref = outs["tile128"][0]
        scale = float(ref.abs().max())
        for tag in ("wide1", "wide2"):
            y, name = outs[tag]
            if not name.startswith("gemm_wide"):
                import warnings
                warnings.warn("nacf_amd: the self-test of the fixed-register GEMM kernels could not select %s (got %r): not checked" % (tag, name))
                continue
            err = float((y - ref).abs().max())
            if not (err <= 1e-5 * max(scale, 1.0)):
                raise RuntimeError("nacf_amd: the %s kernel disagrees with the 128x128 kernel (max |diff| %.3e of %.3e): this "
                                   "libnacf_hip.so was not built by csrc/Makefile's checked recipe -- rebuild it "
                                   "(python __graft_entry__.py)" % (name, err, scale))
    finally:
        set_gemm_mode(mode)
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


class WeightImages:
    """bf16 image planes of the GEMM weight matrices that live in ONE flat fp32 buffer (see nacf_wimage_* in
    nacf_hip.h).  mats: [(offset, N, K, want_transposed)], each matrix contiguous ([N, K], row pitch K) at
    flat[offset:].  Forward images (k-tile-major, zero-padded to whole k-tiles) serve y = x W^T -- also for a run of
    whole rows of a registered matrix -- transposed ones the dX GEMMs.  `refresh()` is one launch."""

    def __init__(self, flat: Tensor, mats, ns: int, _selftest: bool = False):
        assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous() and flat.data_ptr() % 16 == 0
        assert ns in (1, 3)
        if ns == 3 and not _selftest:
            selftest_fixed_register_kernels(flat.device)
        self.flat, self.ns, self.mats = flat, int(ns), list(mats)
        up32 = lambda v: (v + 31) // 32 * 32
        f_off, t_off, f_total, t_total = [], [], 0, 0
        for off, N, K, want_t in self.mats:
            assert off + N * K <= flat.numel()
            f_off.append(f_total)
            f_total += up32(K) * N
            t_off.append(t_total if want_t else None)
            if want_t:
                t_total += up32(N) * K
        self.img = torch.zeros(ns, max(f_total, 8), dtype=torch.int16, device=flat.device)
        self.imgT = torch.zeros(ns, max(t_total, 8), dtype=torch.int16, device=flat.device)
        descs = (L.WImageDesc * max(len(self.mats), 1))()
        tile0 = 0
        lib = L.load()
        for d, (off, N, K, want_t), fo, to in zip(descs, self.mats, f_off, t_off):
            d.w = flat.data_ptr() + 4 * off
            d.img = self.img.data_ptr() + 2 * fo
            d.imgT = self.imgT.data_ptr() + 2 * to if want_t else None
            d.ld, d.plane, d.planeT = K, self.img.stride(0), self.imgT.stride(0)
            d.N, d.K, d.tile0, d.tiles_k = N, K, tile0, (K + 31) // 32
            tile0 += ((N + 31) // 32) * ((K + 31) // 32)
            L.check(lib.nacf_wimage_register(ctypes.c_void_p(d.w), N, K, K, ctypes.c_void_p(d.img), d.plane,
                                             ctypes.c_void_p(d.imgT) if want_t else None, d.planeT, self.ns),
                    "nacf_wimage_register")
        self.n_tiles = tile0
        self.n_desc = len(self.mats)
        self.table = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(flat.device)
        self._registered = True

    def refresh(self) -> None:
        """rebuild every image from the current fp32 values (asynchronous on the current stream; capturable)"""
        if self.n_desc:
            L.check(L.load().nacf_wimage_refresh(_ptr(self.table), self.n_desc, self.n_tiles, self.ns, _stream()),
                    "nacf_wimage_refresh")

    def close(self) -> None:
        if getattr(self, "_registered", False):
            self._registered = False
            try:
                L.load().nacf_wimage_unregister(_ptr(self.flat), self.flat.numel())
            except Exception:  # noqa: BLE001  (interpreter shutdown)
                pass

    def __del__(self):
        self.close()


class RngState:
    """Device-side {seed, step} consumed by every dropout kernel."""

    def __init__(self, seed: int, device):
        self.state = torch.tensor([seed, 0], dtype=torch.int64, device=device)

    def advance(self):
        L.check(L.load().nacf_rng_advance(_ptr(self.state), _stream()), "nacf_rng_advance")


def stacked_rows(parts):
    """torch.cat(parts, dim=0) of equal-shape contiguous tensors -- WITHOUT a copy when they already sit back to back
    in one storage (the step engine allocates the two NACF passes' tokens / labels that way): the decoder batches
    both passes as 2B rows every step, and a copy kernel per step is pure overhead."""
    a = parts[0]
    ok = all(t.is_contiguous() and t.shape == a.shape and t.dtype == a.dtype and t.device == a.device for t in parts)
    if ok:
        step = a.numel() * a.element_size()
        ok = all(t.untyped_storage().data_ptr() == a.untyped_storage().data_ptr() and
                 t.data_ptr() == a.data_ptr() + i * step for i, t in enumerate(parts))
    if ok:
        return torch.as_strided(a, (len(parts) * a.shape[0],) + tuple(a.shape[1:]), a.stride())
    return torch.cat(list(parts), dim=0)


# ---------------------------------------------------------------- live-row lists
class RowSet:
    """Device-side list of the live (non-<pad>) activation slots; see nacf_rowset in nacf_hip.h."""
    __slots__ = ("rows", "count")

    def __init__(self, rows: Tensor, count: Tensor):
        self.rows, self.count = rows, count

    def c(self, zero_dead: bool = False):
        r = L.RowSet()
        r.rows, r.count, r.zero_dead = self.rows.data_ptr(), self.count.data_ptr(), int(zero_dead)
        return ctypes.byref(r)


def _rs(rows: Optional["RowSet"], zero_dead: bool = False):
    return rows.c(zero_dead) if rows is not None else None


def rowset_build(tokens: Optional[Tensor] = None, flags: Optional[Tensor] = None) -> RowSet:
    """slots i with tokens[i] != PAD (and flags[i] != 0), ascending; no host sync"""
    ref = tokens if tokens is not None else flags
    n = ref.numel()
    rows = torch.empty(n, dtype=torch.int32, device=ref.device)
    count = torch.empty(1, dtype=torch.int32, device=ref.device)
    L.check(L.load().nacf_rowset_build(_ptr(tokens), _ptr(flags), n, _ptr(rows), _ptr(count), _stream()),
            "nacf_rowset_build")
    return RowSet(rows, count)


# ---------------------------------------------------------------- linear
class Epi:
    """Python-side description of nacf_epilogue (kept alive through backward)."""
    __slots__ = ("bias", "act", "act_split", "preact", "p1", "salt1", "residual", "p2", "salt2",
                 "row_tokens", "rng")

    def __init__(self, bias=None, act=L.ACT_NONE, act_split=0, preact=None, p1=0.0, salt1=0,
                 residual=None, p2=0.0, salt2=0, row_tokens=None, rng: Optional[RngState] = None):
        self.bias, self.act, self.act_split, self.preact = bias, act, act_split, preact
        self.p1, self.salt1, self.residual, self.p2, self.salt2 = p1, salt1, residual, p2, salt2
        self.row_tokens, self.rng = row_tokens, rng

    def cstruct(self) -> L.Epilogue:
        e = L.Epilogue()
        e.bias = self.bias.data_ptr() if self.bias is not None else None
        e.act, e.act_split = self.act, self.act_split
        if self.preact is not None:
            e.preact, e.ld_preact = self.preact.data_ptr(), self.preact.stride(0)
        e.p_drop1, e.salt1 = float(self.p1), int(self.salt1) & 0xFFFFFFFF
        if self.residual is not None:
            e.residual, e.ld_residual = self.residual.data_ptr(), self.residual.stride(0)
        e.p_drop2, e.salt2 = float(self.p2), int(self.salt2) & 0xFFFFFFFF
        e.row_tokens = self.row_tokens.data_ptr() if self.row_tokens is not None else None
        e.rng_state = self.rng.state.data_ptr() if self.rng is not None else None
        return e


def linear_fwd(x: Tensor, w: Tensor, out: Tensor, epi: Optional[Epi] = None, rows: Optional[RowSet] = None,
               zero_dead: bool = False) -> Tensor:
    """out[M,N] = epilogue(x[M,K] @ w[N,K]^T).  With a row set only the live rows are computed; zero_dead also
    writes zeros to the other rows of `out` (and of epi.preact), so `out` needs no initialisation."""
    _chk_f32(x, w, out)
    M, K, ldx = _rows2d(x)
    N, K2, ldw = _rows2d(w)
    assert K == K2 and out.shape == (M, N), (x.shape, w.shape, out.shape)
    ep = epi.cstruct() if epi is not None else L.Epilogue()
    tok = PROFILER.begin(0, M, N, K, "EpiLinear", rows,
                         heavy=epi is not None and epi.act in (L.ACT_GELU_NEW, L.ACT_GELU_ERF, L.ACT_TANH, L.ACT_SIGMOID,
                                                               L.ACT_TANH_SIGMOID))
    L.check(L.load().nacf_linear_fwd(_ptr(x), ldx, _ptr(w), ldw, _ptr(out), out.stride(0), M, N, K,
                                     ctypes.byref(ep), _rs(rows, zero_dead), _stream()), "nacf_linear_fwd")
    PROFILER.end(tok)
    wide_group.note(M, N, K, rows)
    return out


def linear_bwd_data(dz: Tensor, w: Tensor, dx: Tensor, beta: float = 0.0, rows: Optional[RowSet] = None,
                    zero_dead: bool = False) -> Tensor:
    _chk_f32(dz, w, dx)
    M, N, lddz = _rows2d(dz)
    N2, K, ldw = _rows2d(w)
    assert N == N2 and dx.shape == (M, K)
    lib = L.load()
    ws = _ws().get(lib.nacf_linear_bwd_data_workspace(M, N, K), dz.device)
    tok = PROFILER.begin(1, M, N, K, "EpiStore", rows)
    L.check(lib.nacf_linear_bwd_data(_ptr(dz), lddz, _ptr(w), ldw, _ptr(dx), dx.stride(0), M, N, K,
                                     float(beta), _ptr(ws), ws.numel(), _rs(rows, zero_dead), _stream()),
            "nacf_linear_bwd_data")
    PROFILER.end(tok)
    wide_group.note(M, N, K, rows)
    return dx


class _DwGroup:
    """Deferred split-K combines of the weight-gradient GEMMs of one backward pass (nacf_dw_group_*, include/nacf_hip.h).

    While a group is open every linear_bwd_weight call launches its GEMM at once but leaves the combine of its partial
    slabs for `flush`, which runs them all in ONE launch (12 ~10 us kernels per NACF step otherwise).  Each call gets its
    own slice of a dedicated grow-only buffer (the slabs must survive until the flush); a second gradient for the same dW
    (a layer applied twice) flushes first, so its read-modify-write sees the first one's result."""

    def __init__(self):
        self.ws = _Workspace()
        self.active = False
        self.defer_gemm = False
        self.offset = 0
        self.targets = set()
        self.keep = []           # operands of deferred GEMMs: alive (and unmodified) until the flush has launched them
        self.device = None

    def begin(self, defer_gemm: bool = True):
        if self.active:
            self.flush()
        # the GEMM profiler times one launch per call: keep the GEMMs where they are while it is on
        self.defer_gemm = bool(defer_gemm) and not PROFILER.enabled
        L.check(L.load().nacf_dw_group_begin(int(self.defer_gemm)), "nacf_dw_group_begin")
        self.active, self.offset = True, 0
        self.targets.clear()

    def flush(self):
        if self.active:
            self.active = False
            self.targets.clear()
            self.offset = 0
            try:
                if PROFILER.group_enabled and self.defer_gemm and self.keep:
                    probs = [(dz.shape[0], dz.shape[1], x.shape[1], rows) for dz, x, rows, _dw, _db in self.keep]
                    PROFILER.group_span(probs, lambda: L.check(min(0, L.load().nacf_dw_group_launch_gemms(_stream())),
                                                               "nacf_dw_group_launch_gemms"))
                L.check(L.load().nacf_dw_group_flush(_stream()), "nacf_dw_group_flush")
            finally:
                self.keep.clear()

    def _restart(self):
        d = self.defer_gemm
        self.flush()
        self.begin(d)

    def region(self, need: int, dw: Tensor) -> Tensor:
        key = dw.data_ptr()
        if key in self.targets:
            self._restart()
        need = (int(need) + 255) // 256 * 256
        buf = self.ws.buf.get(dw.device.index if dw.device.index is not None else torch.cuda.current_device())
        if buf is None or self.offset + need > buf.numel():
            # queued combines point into the buffer: run them before it is replaced (the restart rewinds the offset)
            if self.offset > 0:
                self._restart()
            if torch.cuda.is_current_stream_capturing() and buf is not None and need <= buf.numel():
                # under a stream capture growing is an allocation (refused); the rewound buffer holds this region
                self.ws.get(1, dw.device)
            else:
                # outside a capture grow, so that the NEXT pass fits the whole group in one piece (one grouped launch
                # per backward stage instead of one per restart)
                have = 0 if buf is None else buf.numel()
                buf = self.ws.get(max(need, 2 * have, 64 << 20), dw.device)
        elif torch.cuda.is_current_stream_capturing():
            self.ws.get(1, dw.device)          # mark the buffer as seen by a capture (never freed from now on)
        self.targets.add(key)
        out = buf[self.offset:self.offset + need]
        self.offset += need
        return out


DW_GROUP = _DwGroup()


class dw_group:
    """with ops.dw_group(): <a backward pass>  -- combines deferred inside, all run when the block exits"""

    def __init__(self, defer_gemm: bool = True):
        self.defer_gemm = defer_gemm

    def __enter__(self):
        mode = os.environ.get("NACF_DW_GROUP", "2")          # tuning switch: 0 = off, 1 = combines only, 2 = GEMMs too
        if mode != "0":
            DW_GROUP.begin(self.defer_gemm and mode != "1")
        return DW_GROUP

    def __exit__(self, et, ev, tb):
        if et is None:
            DW_GROUP.flush()
        else:                                  # a failed backward: drop the queue, keep the original error
            DW_GROUP.active = False
            L.load().nacf_dw_group_flush(_stream())
            DW_GROUP.keep.clear()
        return False


class wide_group:
    """with ops.wide_group(): <forward / dX GEMMs of INDEPENDENT problems>  -- those the wide-tile kernel serves are queued
    and launched as one grid when the block exits (nacf_wide_group_*, include/nacf_hip.h); everything else launches at
    once.  Inside the block nothing may read what a queued GEMM writes."""

    current = None

    def __enter__(self):
        self.queued = []
        if PROFILER.enabled:             # the per-call profiler times one launch per call: no grouping while it is on
            return self
        L.check(L.load().nacf_wide_group_begin(), "nacf_wide_group_begin")
        wide_group.current = self
        return self

    def __exit__(self, exc_type, exc, tb):
        if wide_group.current is not self:
            return False
        wide_group.current = None
        rc = PROFILER.group_span(self.queued, lambda: L.load().nacf_wide_group_flush(_stream())) if self.queued \
            else L.load().nacf_wide_group_flush(_stream())
        if exc_type is None and rc < 0:
            L.check(rc, "nacf_wide_group_flush")
        return False

    @staticmethod
    def note(M, N, K, rows):
        """called after a forward / dX GEMM entry point: remember the problem if the library queued it"""
        g = wide_group.current
        if g is not None and (L.load().nacf_gemm_last_kernel() or b"") == b"gemm_wide_queued":
            g.queued.append((M, N, K, rows))


def dw_group_begin(defer_gemm: bool = True) -> None:
    DW_GROUP.begin(defer_gemm)


def dw_group_flush() -> None:
    DW_GROUP.flush()


def linear_bwd_weight(dz: Tensor, x: Tensor, dw: Tensor, db: Optional[Tensor], beta: float = 1.0,
                      rows: Optional[RowSet] = None) -> None:
    _chk_f32(dz, x, dw, db)
    M, N, lddz = _rows2d(dz)
    M2, K, ldx = _rows2d(x)
    assert M == M2 and dw.shape == (N, K), (dz.shape, x.shape, dw.shape)
    lib = L.load()
    need = lib.nacf_linear_bwd_weight_workspace(M, N, K)
    ws = DW_GROUP.region(need, dw) if DW_GROUP.active else _ws().get(need, dz.device)
    if DW_GROUP.active and DW_GROUP.defer_gemm:
        DW_GROUP.keep.append((dz, x, rows, dw, db))
    tok = PROFILER.begin(2, M, N, K, "EpiStore", rows)
    L.check(lib.nacf_linear_bwd_weight(_ptr(dz), lddz, _ptr(x), ldx, _ptr(dw), dw.stride(0), _ptr(db), M, N, K,
                                       float(beta), _ptr(ws), ws.numel(), _rs(rows), _stream()),
            "nacf_linear_bwd_weight")
    PROFILER.end(tok)


def loss_combine(slab: Tensor, n_terms: int, stride: int, coef: Tensor, total: Tensor, m_dst: Optional[Tensor],
                 m_src: Optional[Tensor], m_scale: Optional[Tensor], meters: Optional[Tensor]) -> None:
    n_m = 0 if m_dst is None else m_dst.numel()
    L.check(L.load().nacf_loss_combine(_ptr(slab), n_terms, stride, _ptr(coef), _ptr(total), _ptr(m_dst), _ptr(m_src),
                                       _ptr(m_scale), n_m, _ptr(meters), _stream()), "nacf_loss_combine")


def loss_combine_bwd(gtotal: Tensor, coef: Tensor, n_terms: int, stride: int, gslab: Tensor) -> None:
    L.check(L.load().nacf_loss_combine_bwd(_ptr(gtotal), _ptr(coef), n_terms, stride, _ptr(gslab), _stream()),
            "nacf_loss_combine_bwd")


class CritTail:
    """What the producers of the criterion's terms leave for the ONE tail launch (nacf_crit_tail_fwd / _bwd) instead of launching
    their own reductions: per decoding pass (label_logp, argmax, labels, exclude, slot), and the length head's (x, t, slot).
    kl_dx: the length head's gradient, written by the backward tail launch for KLDivMeanFn.backward to hand on."""

    def __init__(self):
        self.passes = []
        self.kl = None
        self.kl_dx = None

    def add_pass(self, label_logp, argmax, labels, exclude, slot):
        assert len(self.passes) < 4
        self.passes.append((label_logp, argmax, labels, bool(exclude), int(slot)))

    def cstruct(self):
        t = L.CritTail()
        t.n_pass = len(self.passes)
        for i, (lp, am, lab, ex, slot) in enumerate(self.passes):
            assert lp.is_contiguous() and am.is_contiguous() and lab.is_contiguous() and lp.numel() == am.numel() == lab.numel()
            t.label_logp[i], t.argmax[i], t.labels[i] = lp.data_ptr(), am.data_ptr(), lab.data_ptr()
            t.rows[i], t.exclude[i], t.slot[i] = lp.numel(), int(ex), slot
        if self.kl is not None:
            x, tt, slot = self.kl
            assert x.is_contiguous() and tt.is_contiguous() and x.numel() == tt.numel()
            t.kl_x, t.kl_t, t.kl_total, t.kl_slot = x.data_ptr(), tt.data_ptr(), x.numel(), slot
        return t


def crit_tail_fwd(tail: CritTail, slab: Tensor, n_terms: int, stride: int, coef: Tensor, total: Tensor, m_dst: Optional[Tensor],
                  m_src: Optional[Tensor], m_scale: Optional[Tensor], meters: Optional[Tensor]) -> None:
    n_m = 0 if m_dst is None else m_dst.numel()
    t = tail.cstruct()
    L.check(L.load().nacf_crit_tail_fwd(ctypes.byref(t), _ptr(slab), n_terms, stride, _ptr(coef), _ptr(total), _ptr(m_dst), _ptr(m_src),
                                        _ptr(m_scale), n_m, _ptr(meters), _stream()), "nacf_crit_tail_fwd")


def crit_tail_bwd(tail: CritTail, gtotal: Tensor, coef: Tensor, n_terms: int, stride: int, gslab: Tensor, kl_dx: Optional[Tensor]) -> None:
    t = tail.cstruct()
    L.check(L.load().nacf_crit_tail_bwd(ctypes.byref(t), _ptr(gtotal), _ptr(coef), n_terms, stride, _ptr(gslab), _ptr(kl_dx), _stream()),
            "nacf_crit_tail_bwd")


def epilogue_bwd(dy: Tensor, dz: Tensor, dr: Optional[Tensor], epi: Epi, accumulate_dr: bool = False,
                 rows: Optional[RowSet] = None) -> None:
    """rows: the forward GEMM's live-row list -- only those rows of dz are written (its consumers take the same list); the
    dead rows of dr get zeros"""
    _chk_f32(dy, dz, dr)
    M, N, lddy = _rows2d(dy)
    ep = epi.cstruct()
    L.check(L.load().nacf_epilogue_bwd(_ptr(dy), lddy, _ptr(dz), dz.stride(0), _ptr(dr),
                                       dr.stride(0) if dr is not None else 0, int(accumulate_dr), M, N,
                                       ctypes.byref(ep), _rs(rows, False), _stream()), "nacf_epilogue_bwd")


# ---------------------------------------------------------------- encoder tail
def highway_mix_fwd(h, tg, out, p, salt, rng):
    _chk_f32(h, tg, out)
    rows, D = h.shape
    L.check(L.load().nacf_highway_mix_fwd(_ptr(h), _ptr(tg), _ptr(out), rows, D, float(p), int(salt),
                                          _ptr(rng.state) if rng else None, _stream()), "nacf_highway_mix_fwd")
    return out


def highway_mix_bwd(dout, h, tg, dh, dp, p, salt, rng):
    _chk_f32(dout, h, tg, dh, dp)
    rows, D = h.shape
    L.check(L.load().nacf_highway_mix_bwd(_ptr(dout), _ptr(h), _ptr(tg), _ptr(dh), _ptr(dp), rows, D, float(p),
                                          int(salt), _ptr(rng.state) if rng else None, _stream()),
            "nacf_highway_mix_bwd")


def bn_concat_fwd(x, out, f_off, weight, bias, running_mean, running_var, nbt, save_mean, save_invstd,
                  training, momentum=0.1, eps=1e-5):
    _chk_f32(x, out, weight, bias, running_mean, running_var, save_mean, save_invstd)
    B, F, D = x.shape
    M_total = out.shape[1]
    lib = L.load()
    ws = WORKSPACE.get(lib.nacf_bn_workspace(B * F, D), x.device)
    L.check(lib.nacf_bn_concat_fwd(_ptr(x), _ptr(out), B, F, D, M_total, f_off, _ptr(weight), _ptr(bias),
                                   _ptr(running_mean), _ptr(running_var), _ptr(nbt), _ptr(save_mean),
                                   _ptr(save_invstd), int(training), float(momentum), float(eps), _ptr(ws),
                                   ws.numel(), _stream()), "nacf_bn_concat_fwd")


def bn_concat_bwd(dout, x, dx, f_off, weight, save_mean, save_invstd, dweight, dbias, beta=1.0):
    _chk_f32(dout, x, dx, weight, save_mean, save_invstd, dweight, dbias)
    B, F, D = x.shape
    M_total = dout.shape[1]
    lib = L.load()
    ws = WORKSPACE.get(lib.nacf_bn_workspace(B * F, D), x.device)
    L.check(lib.nacf_bn_concat_bwd(_ptr(dout), _ptr(x), _ptr(dx), B, F, D, M_total, f_off, _ptr(weight),
                                   _ptr(save_mean), _ptr(save_invstd), _ptr(dweight), _ptr(dbias), float(beta),
                                   _ptr(ws), ws.numel(), _stream()), "nacf_bn_concat_bwd")


def _ptr_array(ts):
    """host array of device pointers (NULL for None) for the *_multi entry points"""
    import ctypes
    return (ctypes.c_void_p * len(ts))(*[(t.data_ptr() if t is not None else None) for t in ts])


def _int_array(vs):
    import ctypes
    return (ctypes.c_int * len(vs))(*[int(v) for v in vs])


def _i64_array(vs):
    import ctypes
    return (ctypes.c_int64 * len(vs))(*[int(v) for v in vs])


def bn_concat_fwd_multi(xs, out, f_offs, weights, biases, running_means, running_vars, nbts, save_means, save_invstds,
                        training, momentum=0.1, eps=1e-5, stats_global=None, n_total=None):
    """every modality of a joint representation in the same three launches (nacf_bn_concat_fwd_multi);
    stats_global [2, n_mod, D] + n_total: the data-parallel form (global-batch statistics, one launch)"""
    _chk_f32(stats_global)
    assert stats_global is None or (stats_global.is_contiguous() and stats_global.shape == (2, len(xs), xs[0].shape[2]))
    _chk_f32(out, *xs, *weights, *biases, *running_means, *running_vars, *save_means, *save_invstds)
    B, _, D = xs[0].shape
    assert all(x.shape[0] == B and x.shape[2] == D and x.is_contiguous() for x in xs)
    lib = L.load()
    n = len(xs)
    ws = WORKSPACE.get(n * lib.nacf_bn_workspace(B * max(x.shape[1] for x in xs), D), out.device)
    L.check(lib.nacf_bn_concat_fwd_multi(n, _ptr_array(xs), _ptr(out), B, _int_array([x.shape[1] for x in xs]), D, out.shape[1],
                                         _int_array(f_offs), _ptr_array(weights), _ptr_array(biases), _ptr_array(running_means),
                                         _ptr_array(running_vars), _ptr_array(nbts), _ptr_array(save_means),
                                         _ptr_array(save_invstds), int(training), float(momentum), float(eps),
                                         _ptr(stats_global), _i64_array(n_total) if n_total is not None else None, _ptr(ws),
                                         ws.numel(), _stream()), "nacf_bn_concat_fwd_multi")


def bn_concat_bwd_multi(dout, xs, dxs, f_offs, weights, save_means, save_invstds, dweights, dbiases, beta=1.0,
                        sums_global=None, n_total=None):
    """sums_global [n_mod, 2, D] + n_total: the data-parallel form (dweights / dbiases untouched)"""
    _chk_f32(dout, sums_global, *xs, *dxs, *weights, *save_means, *save_invstds, *dweights, *dbiases)
    assert sums_global is None or (sums_global.is_contiguous() and sums_global.shape == (len(xs), 2, xs[0].shape[2]))
    B, _, D = xs[0].shape
    lib = L.load()
    n = len(xs)
    ws = WORKSPACE.get(n * lib.nacf_bn_workspace(B * max(x.shape[1] for x in xs), D), dout.device)
    L.check(lib.nacf_bn_concat_bwd_multi(n, _ptr(dout), _ptr_array(xs), _ptr_array(dxs), B, _int_array([x.shape[1] for x in xs]), D,
                                         dout.shape[1], _int_array(f_offs), _ptr_array(weights), _ptr_array(save_means),
                                         _ptr_array(save_invstds), _ptr_array(dweights), _ptr_array(dbiases), float(beta),
                                         _ptr(sums_global), _i64_array(n_total) if n_total is not None else None,
                                         _ptr(ws), ws.numel(), _stream()), "nacf_bn_concat_bwd_multi")


def bn_sync_local_multi(xs, loc):
    """loc [2, n_mod, D] = this rank's (sum | squared deviations about its own mean) per modality (nacf_bn_sync_local_multi)"""
    _chk_f32(loc, *xs)
    B, _, D = xs[0].shape
    n = len(xs)
    assert loc.shape == (2, n, D) and loc.is_contiguous() and all(x.is_contiguous() and x.shape[0] == B and x.shape[2] == D for x in xs)
    lib = L.load()
    ws = WORKSPACE.get(n * lib.nacf_bn_workspace(B * max(x.shape[1] for x in xs), D), loc.device)
    L.check(lib.nacf_bn_sync_local_multi(n, _ptr_array(xs), B, _int_array([x.shape[1] for x in xs]), D, _ptr(loc), _ptr(ws),
                                         ws.numel(), _stream()), "nacf_bn_sync_local_multi")


def bn_sync_bwd_local_multi(dout, xs, f_offs, save_means, save_invstds, sums, dweights, dbiases, beta=1.0):
    """sums [n_mod, 2, D] = this rank's (sum dy | sum dy*xhat); the same sums accumulate into the LOCAL dbiases | dweights"""
    _chk_f32(dout, sums, *xs, *save_means, *save_invstds, *dweights, *dbiases)
    B, _, D = xs[0].shape
    n = len(xs)
    assert sums.shape == (n, 2, D) and sums.is_contiguous()
    lib = L.load()
    ws = WORKSPACE.get(n * lib.nacf_bn_workspace(B * max(x.shape[1] for x in xs), D), dout.device)
    L.check(lib.nacf_bn_sync_bwd_local_multi(n, _ptr(dout), _ptr_array(xs), B, _int_array([x.shape[1] for x in xs]), D, dout.shape[1],
                                             _int_array(f_offs), _ptr_array(save_means), _ptr_array(save_invstds), _ptr(sums),
                                             _ptr_array(dweights), _ptr_array(dbiases), float(beta), _ptr(ws), ws.numel(),
                                             _stream()), "nacf_bn_sync_bwd_local_multi")


def bn_sync_merge(gathered, rows_per_rank, out, ragged_flag=None):
    """gathered [world, 2, n_mod, D] (or flat [world, 2*n_mod*D + n_mod]: every rank's row counts behind its statistics) of per-rank
    (sum | squared deviations about the rank's own mean) -> out [2, n_mod, D] (global sum | squared deviations about the global
    mean); with the counts, a rank holding another number of rows than rows_per_rank turns `out` into NaN and sets
    ragged_flag (device int32[1]); see nacf_bn_sync_merge"""
    import ctypes
    _chk_f32(gathered, out)
    two, n_mod, D = out.shape
    world = gathered.shape[0]
    assert two == 2 and gathered.is_contiguous() and out.is_contiguous()
    stride = gathered.numel() // world
    assert stride in (2 * n_mod * D, 2 * n_mod * D + n_mod), (tuple(gathered.shape), tuple(out.shape))
    rows = (ctypes.c_float * n_mod)(*[float(r) for r in rows_per_rank])
    L.check(L.load().nacf_bn_sync_merge(_ptr(gathered), world, n_mod, D, rows, stride, _ptr(out), _ptr(ragged_flag), _stream()),
            "nacf_bn_sync_merge")


def bn_sync_stat(x, sum_global, n_total, out):
    """one pass of the global-batch BatchNorm statistics over this rank's rows (see nacf_bn_sync_stat): the column
    sums (sum_global None) or the squared deviations from the global mean sum_global / n_total"""
    _chk_f32(x, sum_global, out)
    B, F, D = x.shape
    lib = L.load()
    ws = WORKSPACE.get(lib.nacf_bn_workspace(B * F, D), x.device)
    L.check(lib.nacf_bn_sync_stat(_ptr(x), B * F, D, _ptr(sum_global), int(n_total), _ptr(out), _ptr(ws), ws.numel(), _stream()),
            "nacf_bn_sync_stat")


def bn_concat_fwd_sync(x, out, f_off, weight, bias, running_mean, running_var, nbt, save_mean, save_invstd, sum_global,
                       sqdev_global, n_total, momentum=0.1, eps=1e-5):
    _chk_f32(x, out, weight, bias, running_mean, running_var, save_mean, save_invstd, sum_global, sqdev_global)
    B, F, D = x.shape
    L.check(L.load().nacf_bn_concat_fwd_sync(_ptr(x), _ptr(out), B, F, D, out.shape[1], f_off, _ptr(weight), _ptr(bias),
                                             _ptr(running_mean), _ptr(running_var), _ptr(nbt), _ptr(save_mean),
                                             _ptr(save_invstd), float(momentum), float(eps), _ptr(sum_global),
                                             _ptr(sqdev_global), int(n_total), _stream()), "nacf_bn_concat_fwd_sync")


def bn_sync_bwd_stat(dout, x, f_off, save_mean, save_invstd, sums2, dweight, dbias, beta=1.0):
    _chk_f32(dout, x, save_mean, save_invstd, sums2, dweight, dbias)
    B, F, D = x.shape
    lib = L.load()
    ws = WORKSPACE.get(lib.nacf_bn_workspace(B * F, D), x.device)
    L.check(lib.nacf_bn_sync_bwd_stat(_ptr(dout), _ptr(x), B, F, D, dout.shape[1], f_off, _ptr(save_mean), _ptr(save_invstd),
                                      _ptr(sums2), _ptr(dweight), _ptr(dbias), float(beta), _ptr(ws), ws.numel(), _stream()),
            "nacf_bn_sync_bwd_stat")


def bn_concat_bwd_sync(dout, x, dx, f_off, weight, save_mean, save_invstd, sums2_global, n_total):
    _chk_f32(dout, x, dx, weight, save_mean, save_invstd, sums2_global)
    B, F, D = x.shape
    L.check(L.load().nacf_bn_concat_bwd_sync(_ptr(dout), _ptr(x), _ptr(dx), B, F, D, dout.shape[1], f_off, _ptr(weight),
                                             _ptr(save_mean), _ptr(save_invstd), _ptr(sums2_global), int(n_total), _stream()),
            "nacf_bn_concat_bwd_sync")


def mean_time_fwd(x, out):
    _chk_f32(x, out)
    B, T, D = x.shape
    L.check(L.load().nacf_mean_time_fwd(_ptr(x), _ptr(out), B, T, D, _stream()), "nacf_mean_time_fwd")
    return out


def mean_time_bwd(dout, dx, accumulate=False, dout2=None):
    """dx (+)= (dout + dout2) / T broadcast over time; dout2: the gradient of a second consumer of the mean"""
    _chk_f32(dout, dout2, dx)
    B, T, D = dx.shape
    L.check(L.load().nacf_mean_time_bwd(_ptr(dout), _ptr(dout2), _ptr(dx), B, T, D, int(accumulate), _stream()),
            "nacf_mean_time_bwd")


def log_softmax_rows(x, out):
    _chk_f32(x, out)
    rows, N = x.shape
    L.check(L.load().nacf_log_softmax_rows(_ptr(x), _ptr(out), rows, N, _stream()), "nacf_log_softmax_rows")
    return out


def log_softmax_rows_bwd(dout, out, din):
    _chk_f32(dout, out, din)
    rows, N = out.shape
    L.check(L.load().nacf_log_softmax_rows_bwd(_ptr(dout), _ptr(out), _ptr(din), rows, N, _stream()),
            "nacf_log_softmax_rows_bwd")


def kldiv_mean(x, t, loss_out, dx, gscale=None, scale=1.0):
    _chk_f32(x, t, loss_out, dx, gscale)
    rows, N = x.shape
    L.check(L.load().nacf_kldiv_mean(_ptr(x), _ptr(t), _ptr(loss_out), _ptr(dx), _ptr(gscale), float(scale), rows, N,
                                     _stream()), "nacf_kldiv_mean")


# ---------------------------------------------------------------- decoder
def embed_ln_fwd(tokens, category, additional, word, pos, cat, ln_w, ln_b, out, xhat, rstd, vdiv, vmod, eps,
                 p, salt, rng):
    _chk_f32(additional, word, pos, cat, ln_w, ln_b, out, xhat, rstd)
    R, Lq = tokens.shape
    D = word.shape[1]
    L.check(L.load().nacf_embed_ln_fwd(_ptr(tokens), _ptr(category), _ptr(additional), _ptr(word), _ptr(pos),
                                       _ptr(cat), _ptr(ln_w), _ptr(ln_b), _ptr(out), _ptr(xhat), _ptr(rstd), R, Lq,
                                       D, vdiv, vmod, float(eps), float(p), int(salt),
                                       _ptr(rng.state) if rng else None, _stream()), "nacf_embed_ln_fwd")
    return out


def embed_ln_bwd(dout, xhat, rstd, ln_w, dE, dln_w, dln_b, R, Lq, D, p, salt, rng, beta=1.0):
    _chk_f32(dout, xhat, rstd, ln_w, dE, dln_w, dln_b)
    lib = L.load()
    ws = WORKSPACE.get(lib.nacf_embed_ln_bwd_workspace(R, Lq, D), dout.device)
    L.check(lib.nacf_embed_ln_bwd(_ptr(dout), _ptr(xhat), _ptr(rstd), _ptr(ln_w), _ptr(dE), _ptr(dln_w),
                                  _ptr(dln_b), float(beta), R, Lq, D, float(p), int(salt),
                                  _ptr(rng.state) if rng else None, _ptr(ws), ws.numel(), _stream()),
            "nacf_embed_ln_bwd")


def embed_scatter_bwd(dE, tokens, category, dword, dpos, dcat, dadd, R, Lq, D, V, n_cat, n_video, vdiv, vmod):
    lib = L.load()
    ws = WORKSPACE.get(lib.nacf_embed_scatter_bwd_workspace(R, Lq, D, n_video), dE.device)
    L.check(lib.nacf_embed_scatter_bwd(_ptr(dE), _ptr(tokens), _ptr(category), _ptr(dword), _ptr(dpos), _ptr(dcat),
                                       _ptr(dadd), R, Lq, D, V, n_cat, n_video, vdiv, vmod, _ptr(ws), ws.numel(),
                                       _stream()), "nacf_embed_scatter_bwd")


def layernorm_fwd(x2d, ln_w, ln_b, out, xhat, rstd, eps, seg_in, seg_out, seg_off, p, salt, rng, row_tokens):
    _chk_f32(x2d, ln_w, ln_b, out, xhat, rstd)
    rows, D = x2d.shape
    L.check(L.load().nacf_layernorm_fwd(_ptr(x2d), _ptr(ln_w), _ptr(ln_b), _ptr(out), _ptr(xhat), _ptr(rstd), rows, D,
                                        float(eps), seg_in, seg_out, seg_off, float(p), int(salt),
                                        _ptr(rng.state) if rng else None, _ptr(row_tokens), _stream()),
            "nacf_layernorm_fwd")
    return out


def layernorm_bwd(dout, xhat, rstd, ln_w, dx, dln_w, dln_b, seg_in, seg_out, seg_off, p, salt, rng, row_tokens,
                  beta=1.0):
    _chk_f32(dout, xhat, rstd, ln_w, dx, dln_w, dln_b)
    rows, D = xhat.shape
    lib = L.load()
    ws = WORKSPACE.get(lib.nacf_layernorm_bwd_workspace(rows, D), dout.device)
    L.check(lib.nacf_layernorm_bwd(_ptr(dout), _ptr(xhat), _ptr(rstd), _ptr(ln_w), _ptr(dx), _ptr(dln_w), _ptr(dln_b),
                                   float(beta), rows, D, seg_in, seg_out, seg_off, float(p), int(salt),
                                   _ptr(rng.state) if rng else None, _ptr(row_tokens), _ptr(ws), ws.numel(),
                                   _stream()), "nacf_layernorm_bwd")


def attention_fwd(q, k, v, out, key_tokens, causal, probs, R, H, Lq, Lk, dk, kv_div, kv_mod, drop=None):
    """q/k/v/out are 2-D column slices (rows = seq*len, cols = H*dk) of packed buffers.
    drop = (p, salt, RngState): attention_probs_dropout_prob (models/bert.py:135,169)."""
    _chk_f32(q, k, v, out, probs)
    if drop is not None and drop[0] > 0.0:
        L.check(L.load().nacf_attention_fwd_dropout(_ptr(q), q.stride(0), _ptr(k), k.stride(0), _ptr(v), v.stride(0),
                                                    _ptr(out), out.stride(0), _ptr(key_tokens), int(causal), _ptr(probs),
                                                    R, H, Lq, Lk, dk, kv_div, kv_mod, float(drop[0]), int(drop[1]),
                                                    _ptr(drop[2].state), _stream()), "nacf_attention_fwd_dropout")
        return out
    L.check(L.load().nacf_attention_fwd(_ptr(q), q.stride(0), _ptr(k), k.stride(0), _ptr(v), v.stride(0),
                                        _ptr(out), out.stride(0), _ptr(key_tokens), int(causal), _ptr(probs),
                                        R, H, Lq, Lk, dk, kv_div, kv_mod, _stream()), "nacf_attention_fwd")
    return out


def attention_bwd(q, k, v, do, dq, dk_, dv, key_tokens, causal, R, n_kv, H, Lq, Lk, dk, kv_div, kv_mod, drop=None):
    _chk_f32(q, k, v, do, dq, dk_, dv)
    if drop is not None and drop[0] > 0.0:
        L.check(L.load().nacf_attention_bwd_dropout(_ptr(q), q.stride(0), _ptr(k), k.stride(0), _ptr(v), v.stride(0),
                                                    _ptr(do), do.stride(0), _ptr(dq), dq.stride(0), _ptr(dk_), dk_.stride(0),
                                                    _ptr(dv), dv.stride(0), _ptr(key_tokens), int(causal), R, n_kv, H, Lq, Lk,
                                                    dk, kv_div, kv_mod, float(drop[0]), int(drop[1]), _ptr(drop[2].state),
                                                    _stream()), "nacf_attention_bwd_dropout")
        return
    L.check(L.load().nacf_attention_bwd(_ptr(q), q.stride(0), _ptr(k), k.stride(0), _ptr(v), v.stride(0),
                                        _ptr(do), do.stride(0), _ptr(dq), dq.stride(0), _ptr(dk_), dk_.stride(0),
                                        _ptr(dv), dv.stride(0), _ptr(key_tokens), int(causal), R, n_kv, H, Lq, Lk,
                                        dk, kv_div, kv_mod, _stream()), "nacf_attention_bwd")


def masked_mean_fwd(y, tokens, out):
    _chk_f32(y, out)
    R, Lq, D = y.shape
    L.check(L.load().nacf_masked_mean_fwd(_ptr(y), _ptr(tokens), _ptr(out), R, Lq, D, _stream()),
            "nacf_masked_mean_fwd")
    return out


# ---------------------------------------------------------------- vocabulary / loss
def vocab_ld(V: int) -> int:
    """leading dimension of logits buffers: rows stay 16-byte aligned"""
    return (V + 3) // 4 * 4


def vocab_logsoftmax_fwd(logits2d, V, labels, lse, argmax, label_logp, skip_pad_rows=False):
    _chk_f32(logits2d, lse, label_logp)
    rows = logits2d.shape[0]
    L.check(L.load().nacf_vocab_logsoftmax_fwd(_ptr(logits2d), logits2d.stride(0), rows, V, _ptr(labels), _ptr(lse),
                                               _ptr(argmax), _ptr(label_logp), int(skip_pad_rows), _stream()),
            "nacf_vocab_logsoftmax_fwd")


def vocab_lse_fwd(hidden2d, w, bias, logits2d, labels, lse, argmax, label_logp, rows=None):
    """logits = hidden @ w^T + bias (stored raw) AND lse / argmax / log p(label) per live row, the soft-max statistics
    coming out of the GEMM epilogue (no second pass over [rows, V])"""
    _chk_f32(hidden2d, w, bias, logits2d, lse, label_logp)
    n_rows, K, ldh = _rows2d(hidden2d)
    V = w.shape[0]
    assert logits2d.shape == (n_rows, V) and logits2d.stride(1) == 1
    lib = L.load()
    ws = WORKSPACE.get(lib.nacf_vocab_argmax_workspace(n_rows, V), hidden2d.device)
    tok = PROFILER.begin(0, n_rows, V, K, "EpiArgmax", rows)
    L.check(lib.nacf_vocab_lse_fwd(_ptr(hidden2d), ldh, _ptr(w), w.stride(0), _ptr(bias), n_rows, V, K, _ptr(logits2d),
                                   logits2d.stride(0), _ptr(labels), _ptr(lse), _ptr(argmax), _ptr(label_logp), _ptr(ws),
                                   ws.numel(), _rs(rows), _stream()), "nacf_vocab_lse_fwd")
    PROFILER.end(tok)


def xent_bwd_lse(logits2d, lse, dlogits2d, V, labels, gscale, scale, skip_pad_rows=False):
    rows = logits2d.shape[0]
    L.check(L.load().nacf_xent_bwd_lse(_ptr(logits2d), logits2d.stride(0), _ptr(lse), _ptr(dlogits2d),
                                       dlogits2d.stride(0), rows, V, _ptr(labels), _ptr(gscale), float(scale),
                                       int(skip_pad_rows), _stream()), "nacf_xent_bwd_lse")


def xent_bwd_lse_multi(logits2d, lse, dlogits2d, V, labels, gscales, scale, skip_pad_rows=False):
    """the passes of one [n_pass * rows, V] batch in one launch, pass i scaled by gscales[i][0] (nacf_xent_bwd_lse_multi)"""
    n = len(gscales)
    rows = logits2d.shape[0]
    assert rows % n == 0
    L.check(L.load().nacf_xent_bwd_lse_multi(_ptr(logits2d), logits2d.stride(0), _ptr(lse), _ptr(dlogits2d), dlogits2d.stride(0),
                                             rows // n, n, V, _ptr(labels), _ptr_array(gscales), float(scale), int(skip_pad_rows),
                                             _stream()), "nacf_xent_bwd_lse_multi")


def nll_reduce_multi(label_logp, argmax, labels, exclude_masks, outs):
    """per-pass criterion scalars of n_pass passes back to back, one launch (nacf_nll_reduce_multi)"""
    n = len(outs)
    rows = label_logp.numel()
    assert rows % n == 0 and len(exclude_masks) == n
    L.check(L.load().nacf_nll_reduce_multi(_ptr(label_logp), _ptr(argmax), _ptr(labels), rows // n, n,
                                           _int_array([int(bool(e)) for e in exclude_masks]), _ptr_array(outs), _stream()),
            "nacf_nll_reduce_multi")


def nll_reduce(label_logp, argmax, labels, exclude_mask, out5):
    rows = label_logp.numel()
    L.check(L.load().nacf_nll_reduce(_ptr(label_logp), _ptr(argmax), _ptr(labels), rows, int(exclude_mask),
                                     _ptr(out5), _stream()), "nacf_nll_reduce")


def xent_bwd(logp2d, dlogits2d, V, labels, gscale, scale, skip_pad_rows=False):
    rows = logp2d.shape[0]
    L.check(L.load().nacf_xent_bwd(_ptr(logp2d), logp2d.stride(0), _ptr(dlogits2d), dlogits2d.stride(0), rows, V,
                                   _ptr(labels), _ptr(gscale), float(scale), int(skip_pad_rows), _stream()),
            "nacf_xent_bwd")


def vocab_logsoftmax_bwd(dlogp2d, logp2d, dlogits2d, V):
    rows = logp2d.shape[0]
    L.check(L.load().nacf_vocab_logsoftmax_bwd(_ptr(dlogp2d), dlogp2d.stride(0), _ptr(logp2d), logp2d.stride(0),
                                               _ptr(dlogits2d), dlogits2d.stride(0), rows, V, _stream()),
            "nacf_vocab_logsoftmax_bwd")


# ---------------------------------------------------------------- NA decoding
def vocab_argmax(hidden2d, w, bias, pad_tokens, zero_mask_prob, update_mask, tokens, probs, rows=None):
    _chk_f32(hidden2d, w, bias, probs)
    n_rows, K, ldh = _rows2d(hidden2d)
    V = w.shape[0]
    lib = L.load()
    ws = WORKSPACE.get(lib.nacf_vocab_argmax_workspace(n_rows, V), hidden2d.device)
    tok = PROFILER.begin(0, n_rows, V, K, "EpiArgmax", rows)
    L.check(lib.nacf_vocab_argmax(_ptr(hidden2d), ldh, _ptr(w), w.stride(0), _ptr(bias), n_rows, V, K,
                                  _ptr(pad_tokens), int(zero_mask_prob), _ptr(update_mask), _ptr(tokens),
                                  _ptr(probs), _ptr(ws), ws.numel(), _rs(rows), _stream()), "nacf_vocab_argmax")
    PROFILER.end(tok)


def length_beam(pred_length, lbs, length_bias, beam, beam_max):
    _chk_f32(pred_length)
    B, max_len = pred_length.shape
    L.check(L.load().nacf_length_beam(_ptr(pred_length), B, max_len, lbs, length_bias, _ptr(beam), _ptr(beam_max),
                                      _stream()), "nacf_length_beam")


def length_beam_gold(tgt_tokens, max_len, lbs, beam, beam_max):
    assert tgt_tokens.dtype == torch.int64 and tgt_tokens.dim() == 2 and tgt_tokens.is_contiguous()
    B, T = tgt_tokens.shape
    L.check(L.load().nacf_length_beam_gold(_ptr(tgt_tokens), B, T, max_len, lbs, _ptr(beam), _ptr(beam_max), _stream()),
            "nacf_length_beam_gold")


def canvas_init(beam, rows, Lp, tokens):
    L.check(L.load().nacf_canvas_init(_ptr(beam), rows, Lp, _ptr(tokens), _stream()), "nacf_canvas_init")


def canvas_init_gold(beam, tgt_tokens, rows, lbs, Lp, tokens):
    assert tgt_tokens.dtype == torch.int64 and tgt_tokens.is_contiguous() and tgt_tokens.shape[0] * lbs == rows
    L.check(L.load().nacf_canvas_init_gold(_ptr(beam), _ptr(tgt_tokens), tgt_tokens.shape[1], rows, lbs, Lp, _ptr(tokens),
                                           _stream()), "nacf_canvas_init_gold")


def select_mask(probs, teacher, pad_tokens, lut, mode, tokens, mask_out):
    rows, Lp = tokens.shape
    L.check(L.load().nacf_select_mask(_ptr(probs), _ptr(teacher), _ptr(pad_tokens), _ptr(lut), mode, rows, Lp,
                                      _ptr(tokens), _ptr(mask_out), _stream()), "nacf_select_mask")


def token_replace(tokens, src, dst):
    L.check(L.load().nacf_token_replace(_ptr(tokens), tokens.numel(), int(src), int(dst), _stream()),
            "nacf_token_replace")


def teacher_probs(label_logp, pad_tokens, out):
    L.check(L.load().nacf_teacher_probs(_ptr(label_logp), _ptr(pad_tokens), _ptr(out), out.numel(), _stream()),
            "nacf_teacher_probs")


def init_probs(pad_tokens, probs):
    L.check(L.load().nacf_init_probs(_ptr(pad_tokens), _ptr(probs), probs.numel(), _stream()), "nacf_init_probs")


def apply_mask(tokens, mask, value):
    L.check(L.load().nacf_apply_mask(_ptr(tokens), _ptr(mask), int(value), tokens.numel(), _stream()),
            "nacf_apply_mask")


def mask_rank(tokens, rank, counts):
    rows, Lp = tokens.shape
    L.check(L.load().nacf_mask_rank(_ptr(tokens), rows, Lp, _ptr(rank), _ptr(counts), _stream()), "nacf_mask_rank")


def select_rank(rank, cur, q, tokens, mask_out):
    rows, Lp = tokens.shape
    L.check(L.load().nacf_select_rank(_ptr(rank), int(cur), int(q), rows, Lp, _ptr(tokens), _ptr(mask_out),
                                      _stream()), "nacf_select_rank")


def easy_first_update(tokens, probs, new_tokens, new_probs, q):
    rows, Lp = tokens.shape
    L.check(L.load().nacf_easy_first_update(_ptr(tokens), _ptr(probs), _ptr(new_tokens), _ptr(new_probs), int(q),
                                            rows, Lp, _stream()), "nacf_easy_first_update")


def best_candidate(tokens, probs, teacher, beam, alpha, B, lbs, Lp, out_tokens, best_idx, cand_lprobs):
    L.check(L.load().nacf_best_candidate(_ptr(tokens), _ptr(probs), _ptr(teacher), _ptr(beam), float(alpha), B, lbs,
                                         Lp, _ptr(out_tokens), _ptr(best_idx), _ptr(cand_lprobs), _stream()),
            "nacf_best_candidate")


# ---------------------------------------------------------------- optimiser
def adam_step(param, grad, m, v, lr_dev, step_dev, beta1, beta2, eps, weight_decay, grad_clip, grad_scale, bump=True,
              zero_grad=False):
    """param / grad / m / v may be equal-length slices of the flat buffers; bump: this call starts a new step;
    zero_grad: the gradient slice is left zeroed (the step engine's next zero_grad, folded into this walk)"""
    _chk_f32(param, grad, m, v, lr_dev)
    assert param.numel() == grad.numel() == m.numel() == v.numel() and param.is_contiguous()
    L.check(L.load().nacf_adam_step_part(_ptr(param), _ptr(grad), _ptr(m), _ptr(v), param.numel(), _ptr(lr_dev),
                                         _ptr(step_dev), float(beta1), float(beta2), float(eps), float(weight_decay),
                                         float(grad_clip), float(grad_scale), int(bool(bump)) | (2 if zero_grad else 0), _stream()),
            "nacf_adam_step")


def rmsprop_step(param, grad, sq, lr_dev, alpha, eps, weight_decay, grad_clip, grad_scale, zero_grad=False):
    """torch.optim.RMSprop (momentum 0, not centred) over equal-length slices of the flat buffers (nacf_rmsprop_step)"""
    _chk_f32(param, grad, sq, lr_dev)
    assert param.numel() == grad.numel() == sq.numel() and param.is_contiguous()
    L.check(L.load().nacf_rmsprop_step(_ptr(param), _ptr(grad), _ptr(sq), param.numel(), _ptr(lr_dev), float(alpha), float(eps),
                                       float(weight_decay), float(grad_clip), float(grad_scale), int(bool(zero_grad)), _stream()),
            "nacf_rmsprop_step")


# ---------------------------------------------------------------- AR beam search
def beam_step(logp2d, V, t, max_len, want, seqs, scores, fin_scores, fin_len, fin_tokens, fin_count, done, n_active):
    _chk_f32(logp2d, scores, fin_scores)
    B, n_bm = scores.shape
    L.check(L.load().nacf_beam_step(_ptr(logp2d), logp2d.stride(0), B, n_bm, V, int(t), int(max_len), int(want),
                                    _ptr(seqs), _ptr(scores), _ptr(fin_scores), _ptr(fin_len), _ptr(fin_tokens),
                                    _ptr(fin_count), _ptr(done), _ptr(n_active), _stream()), "nacf_beam_step")


# ---------------------------------------------------------------- batch construction
def sample_frames(src: Tensor, video: Optional[Tensor], src_len: Optional[Tensor], n_frames: int, mode: int,
                  out: Tensor, salt: int = 0, rng: Optional[RngState] = None, frame_ids: Optional[Tensor] = None) -> Tensor:
    """out[b, i] = src[video[b], frame(b, i)]; mode 0 equally_sampling, 1 segment_random (dataloader.py:24-37).
    `src` may be a PINNED host tensor: pinned memory is mapped into the device's address space, so the kernel pulls
    exactly the sampled rows over PCIe (zero-copy gather) -- no staging copy of whole clips."""
    if src.is_cuda:
        _chk_f32(src, out)
    else:
        if not src.is_pinned() or src.dtype != torch.float32:
            raise L.NacfLibraryError("nacf_sample_frames: a host source must be pinned float32 memory")
        _chk_f32(out)
    B = out.shape[0]
    T, D = src.shape[-2], src.shape[-1]
    assert src.is_contiguous() and out.is_contiguous() and out.shape == (B, n_frames, D)
    L.check(L.load().nacf_sample_frames(_ptr(src), _ptr(video), _ptr(src_len), B, T, D, n_frames, int(mode),
                                        int(salt) & 0xFFFFFFFF, _ptr(rng.state) if rng else None, _ptr(out),
                                        _ptr(frame_ids), _stream()), "nacf_sample_frames")
    return out


def gather_clips_h2d(dst: Tensor, src_host: Tensor, rows_host) -> None:
    """dst[j] = src_host[rows_host[j]] (whole clips), one asynchronous DMA per clip issued from C on the current
    stream; src_host: pinned [N, T, D] tensor, rows_host: int32 numpy array"""
    import numpy as np
    rows_host = np.ascontiguousarray(rows_host, dtype=np.int32)
    n = int(rows_host.shape[0])
    assert dst.is_cuda and dst.is_contiguous() and src_host.is_pinned() and src_host.is_contiguous()
    assert dst.shape[0] >= n and dst.shape[1:] == src_host.shape[1:] and dst.dtype == src_host.dtype
    assert n == 0 or (0 <= int(rows_host.min()) and int(rows_host.max()) < src_host.shape[0])
    clip = src_host[0].numel() * src_host.element_size()
    L.check(L.load().nacf_gather_clips_h2d(_ptr(dst), _ptr(src_host), ctypes.c_void_p(rows_host.ctypes.data), n, clip,
                                           _stream()), "nacf_gather_clips_h2d")


def gather_clips_zc(dst: Tensor, src_host: Tensor, rows_dev: Tensor, workgroups: int = 0) -> None:
    """dst[j] = src_host[rows_dev[j]] (whole clips) by kernel-issued PCIe reads on the current stream; src_host: pinned
    [N, T, D] tensor (device-mapped), rows_dev: int32 device tensor"""
    n = int(rows_dev.numel())
    assert dst.is_cuda and dst.is_contiguous() and src_host.is_pinned() and src_host.is_contiguous()
    assert rows_dev.is_cuda and rows_dev.dtype == torch.int32 and rows_dev.is_contiguous()
    assert dst.shape[0] >= n and dst.shape[1:] == src_host.shape[1:] and dst.dtype == src_host.dtype
    clip = src_host[0].numel() * src_host.element_size()
    L.check(L.load().nacf_gather_clips_zc(_ptr(dst), ctypes.c_void_p(src_host.data_ptr()), _ptr(rows_dev), n, clip, int(workgroups),
                                          _stream()), "nacf_gather_clips_zc")


def build_targets(caps: Tensor, cap_len: Tensor, pos_tags: Optional[Tensor], tag_demanded: Optional[Tensor],
                  word_is_be: Optional[Tensor], max_len: int, narformer: bool, visual_word: bool, train: bool,
                  beta=(0.0, 1.0), salt: int = 0, rng: Optional[RngState] = None, into=None):
    """decoder inputs / labels of a batch of captions (dataloader.py:317-425) -> dict of int64 [B, max_len];
    `into`: optional dict of preallocated output tensors of that shape (used when they fit)"""
    assert caps.dtype == torch.int32 and caps.dim() == 2 and caps.stride(1) == 1 and cap_len.dtype == torch.int32
    B = caps.shape[0]

    def mk(name):
        t = into.get(name) if into is not None else None
        if torch.is_tensor(t) and t.shape == (B, max_len) and t.dtype == torch.int64 and t.is_contiguous() and t.device == caps.device:
            return t
        return torch.empty(B, max_len, dtype=torch.int64, device=caps.device)
    out = {"tokens": mk("tokens"), "labels": mk("labels")}
    vw = bool(visual_word and train)
    if vw:
        out["tokens_1"], out["labels_1"] = mk("tokens_1"), mk("labels_1")
    L.check(L.load().nacf_build_targets(_ptr(caps), caps.stride(0), _ptr(cap_len), _ptr(pos_tags), _ptr(tag_demanded),
                                        _ptr(word_is_be), B, max_len, int(narformer), int(vw), int(train),
                                        float(beta[0]), float(beta[1]), int(salt) & 0xFFFFFFFF,
                                        _ptr(rng.state) if rng else None, _ptr(out["tokens"]), _ptr(out["labels"]),
                                        _ptr(out.get("tokens_1")), _ptr(out.get("labels_1")), _stream()),
            "nacf_build_targets")
    return out
