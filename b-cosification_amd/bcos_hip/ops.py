"""Tensor-level wrappers over the C ABI: torch is used only for device memory and streams.

All activation tensors here are *physically* NHWC float32 HIP tensors of shape [N,H,W,C]
(i.e. `x_nchw_channels_last.permute(0,2,3,1)`), weights are [Cout,kh,kw,Cin].
"""
import ctypes as C
from typing import Optional

import os

import torch

from . import lib as _l
from .lib import BCOS_CONV_EPS, BCOS_LINEAR_EPS, BCOS_NONE, BcosHipError, Epilogue, TapconvGeom


# bench.py sets this to a list to collect (start, end) HIP events recorded on the launch stream around every
# contraction launch (the roofline's live per-kernel timing); None = off, no overhead.
KERNEL_TIMING = None


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """handle of torch's current stream on the current device (the raw accessor: torch.cuda.current_stream() builds a Stream object
    per call, ~8 us -- a tenth of the host time of a launch-bound training step)"""
    if _RAW_STREAM is not None and _GET_DEVICE is not None:
        return C.c_void_p(_RAW_STREAM(_GET_DEVICE()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t: Optional[torch.Tensor], what: str, contiguous: bool = True):
    """Device pointer of `t` (None -> NULL) after checking it is something the kernels accept."""
    if t is None:
        return None
    if not t.is_cuda:
        raise BcosHipError(
            f"{what}: expected a HIP device tensor, got device={t.device}. The B-cos hot path has no "
            "CPU implementation in this package (the CPU restatement is oracle/, test-only).")
    if t.dtype != torch.float32:
        raise BcosHipError(f"{what}: expected float32, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        raise BcosHipError(f"{what}: expected a contiguous tensor, got strides {t.stride()}")
    return C.c_void_p(t.data_ptr())


def require_device(t: torch.Tensor, who: str = "bcos_hip"):
    """Fail loudly on anything that is not an fp32 HIP tensor: there is no CPU fallback."""
    if not t.is_cuda:
        raise BcosHipError(f"{who}: input is on {t.device}; the B-cos hot path only exists as HIP kernels for "
                           "gfx950 (no CPU fallback) -- move the model and the input to 'cuda'")
    if t.dtype != torch.float32:
        raise BcosHipError(f"{who}: the hot path is fp32 end to end (got {t.dtype})")


def conv_out_size(size, k, s, p, d=1):
    return (size + 2 * p - d * (k - 1) - 1) // s + 1


IMAGE_RANGE_PARTS = os.environ.get("BCOS_IMAGE_RANGE_PARTS", "1") != "0"    # development A/B: 0 = per-image ranges by one workgroup per image (round 5)
FUSE_PATCH_NORM = os.environ.get("BCOS_FUSE_PATCH_NORM", "1") != "0"    # development A/B: 0 = the patch-norm term of pointwise layers as a pass of its own (round 5)
_NO_PRESPLIT = bool(os.environ.get("BCOS_NO_PRESPLIT"))     # development switch: always split inside the kernel
_NO_GROUP = bool(os.environ.get("BCOS_NO_GROUP"))           # development switch: one launch per parity class
_NO_D2S = bool(os.environ.get("BCOS_NO_D2S"))               # development switch: narrow strided gradients on the grouped direct kernel


def mark_static(w: torch.Tensor, transient: bool = False) -> torch.Tensor:
    """Declare `w` an inference-time constant (a layer's effective weight): tapconv() then keeps its pre-split images
    (bcos_split_weights / bcos_split_weights_f16x2) next to it.  In-place updates are noticed through the tensor version
    counter; the images are dropped with the tensor.  `transient`: the weight is being trained -- the images live for one step on the
    stream that made them and are not published to other streams (publish_cached: a device synchronisation per image)."""
    w._bcos_static = True
    if transient and not _PUBLISH_ALWAYS:
        w._bcos_transient = True
    return w


def split_weights(w: torch.Tensor) -> torch.Tensor:
    """Exact 3-way bf16 split of w [rows, ...] in MFMA fragment order (include/bcos_hip.h: bcos_split_weights)."""
    lib = _l.load()
    rows = w.shape[0]
    ktot = w.numel() // rows
    nbytes = C.c_int64(0)
    _l.check(lib.bcos_split_weights_bytes(rows, ktot, C.byref(nbytes)), "bcos_split_weights_bytes")
    out = torch.empty(nbytes.value, device=w.device, dtype=torch.uint8)
    _l.check(lib.bcos_split_weights(_dev(w, "split_weights.w"), C.c_void_p(out.data_ptr()), rows, ktot, _stream()),
             "bcos_split_weights")
    return out


def split_weights_f16x2(w: torch.Tensor, taps: int = 1) -> torch.Tensor:
    """Row-scaled 2-way fp16 split of w [rows, taps * C] in MFMA fragment order + inverse row scales, stored in the K order
    the kernel walks for a launch with `taps` taps (include/bcos_hip.h: bcos_split_weights_f16x2_conv)."""
    lib = _l.load()
    rows = w.shape[0]
    ktot = w.numel() // rows
    if ktot % taps:
        raise BcosHipError(f"split_weights_f16x2: K = {ktot} is not a multiple of taps = {taps}")
    nbytes = C.c_int64(0)
    _l.check(lib.bcos_split_weights_f16x2_bytes(rows, ktot, C.byref(nbytes)), "bcos_split_weights_f16x2_bytes")
    out = torch.empty(nbytes.value, device=w.device, dtype=torch.uint8)
    _l.check(lib.bcos_split_weights_f16x2_conv(_dev(w, "split_weights_f16x2.w"), C.c_void_p(out.data_ptr()), rows, taps, ktot // taps,
                                               _stream()), "bcos_split_weights_f16x2_conv")
    return out


import threading

_TLS = threading.local()          # per thread: a training pass (autograd runs the backward on its own thread) must not switch off the
                                  # publication of images another thread's inference plan is making at the same moment
_PUBLISH_ALWAYS = os.environ.get("BCOS_PUBLISH_ALWAYS", "0") == "1"       # development A/B: every weight image is published (round-4 behaviour)


class transient_weights:
    """Context of the training plans: weight images made inside live for ONE pass and are read on the stream that made them, so
    `publish_cached` has nothing to complete -- its device synchronisation per image (about a hundred per training step) kept the
    host from running ahead of the device: issue time == device time in scripts/probe/train_host_probe.py."""

    def __enter__(self):
        self._prev = getattr(_TLS, "publish", True)
        _TLS.publish = False

    def __exit__(self, *exc):
        _TLS.publish = self._prev
        return False


_PENDING = set()                  # streams on which objects were cached WITHOUT publication (inside a training pass: transient_weights)
_PENDING_LOCK = threading.Lock()


def note_unpublished(t: torch.Tensor):
    """`t` has just been made on the current stream and cached without the synchronisation publish_cached performs (a training pass).
    Remember the stream: the first consumer outside a training pass completes it (publish_pending)."""
    if t is not None and t.is_cuda and not torch.cuda.is_current_stream_capturing():
        with _PENDING_LOCK:
            _PENDING.add(torch.cuda.current_stream(t.device))


def publish_pending():
    """Complete everything that was cached unpublished (engine constants, positional embeddings, weight images of a weight
    version first used under transient_weights and used again, unchanged, by an inference pass on other streams).  Called where
    cached objects are about to be read outside a training pass: one set lookup when nothing is pending."""
    if not _PENDING or not getattr(_TLS, "publish", True) or (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
        return
    with _PENDING_LOCK:
        streams = list(_PENDING)
        _PENDING.clear()
    for st in streams:
        st.synchronize()


def publish_cached(t: torch.Tensor):
    """A device object that has just been made on the CURRENT stream and is about to be cached for later launches, whichever stream
    those run on (the engines process sub-batches on side streams): complete it first.  Once per cached object.  Inside a training
    pass (transient_weights) the synchronisation is deferred to the first reader outside one (publish_pending)."""
    if t is None or not t.is_cuda or torch.cuda.is_current_stream_capturing():
        return
    if getattr(_TLS, "publish", True):
        torch.cuda.current_stream(t.device).synchronize()
    else:
        note_unpublished(t)


def _image_of(wt: torch.Tensor, attr: str, make):
    cached = getattr(wt, attr, None)
    if cached is None or cached[0] != wt._version:
        cached = (wt._version, make(wt))
        setattr(wt, attr, cached)
        if not getattr(wt, "_bcos_transient", False):
            publish_cached(cached[1])
        else:
            note_unpublished(cached[1])
    elif _PENDING:
        publish_pending()          # (an image of this version may have been made, unpublished, by a training pass)
    return cached[1]


# ---- per-pixel max |x| side tensors (the operand scales of the f16x2 contraction) -------------------------------------
class AbsmaxArena:
    """One zero-filled int32 buffer per pass instead of one memset launch per tensor: `reset()` zeroes what the previous
    pass handed out, `take(n)` returns the next n words (torch.zeros once the arena is exhausted; it then grows)."""

    def __init__(self):
        self.buf = None
        self.used = 0
        self.want = 0
        self.gen = 0        # generation: slices handed out before the latest reset() are stale (absmax_of refuses them)

    def reset(self, device):
        self.gen += 1
        need = max(self.want, self.used)
        if self.buf is None or self.buf.device != torch.device(device) or self.buf.numel() < need:
            self.buf = torch.zeros(max(need, 1 << 20), device=device, dtype=torch.int32)
        elif self.used:
            self.buf[:self.used].zero_()
        self.used = self.want = 0

    def take(self, n: int, device):
        n_al = (n + 3) & ~3
        self.want += n_al
        if self.buf is None or self.buf.device != torch.device(device) or self.used + n_al > self.buf.numel():
            return torch.zeros(n, device=device, dtype=torch.int32)
        out = self.buf[self.used:self.used + n]
        out._bcos_arena = (self, self.gen)
        self.used += n_al
        return out


_ARENA: Optional[AbsmaxArena] = None
DEFAULT_TRACK_ABSMAX = True    # what tapconv(track_absmax=None) means while the f16x2 contraction is selected (see no_absmax)
F16X2_MIN_K = int(os.environ.get("BCOS_F16X2_MIN_K", "64"))     # below this K a launch keeps the bf16x3 loop (no operand maxima needed); same-node A/B on ResNet-50: 512 -> 256 = -0.4 ms per step (the K = 256 layers at 14^2 spend a third of their SIMD cycles on the 6 bf16 products); with the LDS-DMA loop of round 3 the K = 64 / 128 layers gain too (256 -> 128: -0.33 ms, -> 64: -0.39 ms per step)


class no_absmax:
    """Context: launches inside do not emit per-pixel maxima unless asked to, so their readers keep the bf16x3 loop.  Used by
    plans whose contractions are too small-K for the f16x2 loop to pay (SimpleViT: K = 192 / 768 linears -- measured
    15.2 k images/s with bf16x3 against 14.5 k with maxima + f16x2 at batch 512)."""

    def __enter__(self):
        global DEFAULT_TRACK_ABSMAX
        self._prev = DEFAULT_TRACK_ABSMAX
        DEFAULT_TRACK_ABSMAX = False

    def __exit__(self, *exc):
        global DEFAULT_TRACK_ABSMAX
        DEFAULT_TRACK_ABSMAX = self._prev
        return False


def set_absmax_arena(arena: Optional[AbsmaxArena]):
    global _ARENA
    _ARENA = arena


class absmax_arena:
    """Context: the launches inside draw their per-pixel maxima from `arena` (reset on entry); the previous arena -- normally
    none: the module / autograd path allocates per tensor -- is restored on exit, so an engine's arena never serves launches
    outside its own pass.  Slices still attached to tensors that outlive the pass go stale at the arena's next reset."""

    def __init__(self, arena: AbsmaxArena, device):
        self.arena, self.device = arena, device

    def __enter__(self):
        self._prev = _ARENA
        self.arena.reset(self.device)
        set_absmax_arena(self.arena)
        return self.arena

    def __exit__(self, *exc):
        set_absmax_arena(self._prev)
        return False


def _new_absmax(n: int, device) -> torch.Tensor:
    return _ARENA.take(n, device) if _ARENA is not None else torch.zeros(n, device=device, dtype=torch.int32)


def _attach_absmax(t: torch.Tensor, am: torch.Tensor):
    t._bcos_absmax = (am, t._version)


def absmax_of(t: torch.Tensor) -> Optional[torch.Tensor]:
    """The valid per-pixel max |t| side tensor of `t`, or None (never produced, or `t` was modified in place since)."""
    rec = getattr(t, "_bcos_absmax", None)
    if rec is None or rec[1] != t._version:
        return None
    src = getattr(rec[0], "_bcos_arena", None)
    if src is not None and src[0].gen != src[1]:      # a slice of an arena that has been reset (and re-issued) since
        return None
    return rec[0]


PATCH_LOOP = os.environ.get("BCOS_PATCH", "1") != "0"     # development / test switch: 3 x 3 launches keep the per-tap loops


FUSE_IMAGE_RANGE = os.environ.get("BCOS_FUSE_IMAGE_RANGE", "1") != "0"     # development / test switch: 0 = always the separate bcos_image_absrange pass
_IMAGE_RANGE_READER = False


class image_range_reader:
    """Context of the engines: the tensor(s) the launches inside write will be read by a launch that takes one operand scale per IMAGE
    (a 3 x 3 / stride-1 layer on the input-patch loop) -- their producers fold the per-image range of the maxima they emit into their own
    epilogue (tapconv: bcos_epilogue.out_imgmax) instead of leaving it to a bcos_image_absrange pass ahead of the reader.  A hint: without
    it (or where a producer cannot fold) the reader computes the range itself; with it and no such reader, a few atomics are wasted."""

    def __init__(self, on: bool = True):
        self.on = bool(on)

    def __enter__(self):
        global _IMAGE_RANGE_READER
        self._prev = _IMAGE_RANGE_READER
        _IMAGE_RANGE_READER = self.on
        return self

    def __exit__(self, *exc):
        global _IMAGE_RANGE_READER
        _IMAGE_RANGE_READER = self._prev
        return False


def reads_image_range(kernel, stride=(1, 1), dilation=(1, 1), groups=1) -> bool:
    """Does a convolution of this geometry take per-image operand scales (the patchable 3 x 3 of tapconv below)?"""
    return tuple(kernel) == (3, 3) and tuple(stride) == (1, 1) and tuple(dilation) == (1, 1) and int(groups) == 1


def image_absmax(am: torch.Tensor, n_images: int, pixels_per_image: int):
    """Per-image range [2, N] of a per-pixel maxima tensor, cached on it -> (range, fused).  fused = False: row 0 maxima, row 1 minima
    over the nonzero pixels, computed here by a pass of its own (include/bcos_hip.h: bcos_image_absrange) -- the maxima of a tensor are
    complete once its producer has been enqueued; the cache is dropped whenever the side tensor is handed to a producer again
    (_out_absmax).  fused = True: the producing launches folded the range into the arrays themselves (bcos_epilogue.out_imgmax /
    out_imgmin_c, tapconv below): row 1 is a complemented lower bound of the minima (bcos_operands.a_imgmin_c)."""
    rec = getattr(am, "_bcos_imgmax", None)
    if rec is not None and rec[1] == (n_images, pixels_per_image):
        return rec[0], len(rec) > 2
    if IMAGE_RANGE_PARTS and _ARENA is not None and pixels_per_image >= 2048:
        # several workgroups per image, meeting by atomics in zero-filled words of the pass's arena (the form the fused epilogues leave:
        # complemented minima): the one-workgroup-per-image kernel is a chain of dependent loads, 30 us per call beside another stream's launch
        out = _new_absmax(2 * n_images, am.device).view(2, n_images)
        _l.check(_l.load().bcos_image_absrange_c(am.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), n_images, pixels_per_image, _stream()),
                 "bcos_image_absrange_c")
        am._bcos_imgmax = (out, (n_images, pixels_per_image), "parts")
        return out, True
    out = torch.empty(2, n_images, device=am.device, dtype=torch.int32)
    _l.check(_l.load().bcos_image_absrange(am.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), n_images, pixels_per_image, _stream()),
             "bcos_image_absrange")
    am._bcos_imgmax = (out, (n_images, pixels_per_image))
    return out, False


def drop_absmax(t: torch.Tensor):
    if hasattr(t, "_bcos_absmax"):
        del t._bcos_absmax


def ensure_absmax(t: torch.Tensor) -> torch.Tensor:
    """Attach the per-pixel max |t| of an NHWC / [rows, C] tensor made by something other than a tapconv epilogue
    (include/bcos_hip.h: bcos_rows_absmax; one extra read of t).  No-op unless the f16x2 contraction is selected."""
    if _l.get_contraction_mode() != "f16x2" or absmax_of(t) is not None:
        return t
    Cc = t.shape[-1]
    if Cc % 4 != 0 or not t.is_contiguous():
        return t
    rows = t.numel() // Cc
    am = torch.empty(rows, device=t.device, dtype=torch.int32)
    _l.check(_l.load().bcos_rows_absmax(_dev(t, "rows_absmax.x"), C.c_void_p(am.data_ptr()), rows, Cc, Cc, _stream()),
             "bcos_rows_absmax")
    _attach_absmax(t, am)
    return t


def _out_absmax(t: Optional[torch.Tensor], pixels: int):
    """absmax buffer of an output tensor: reused when several launches fill disjoint pixels of the same tensor object."""
    if t is None:
        return None
    am = absmax_of(t)
    if am is None or am.numel() != pixels:
        am = _new_absmax(pixels, t.device)
        _attach_absmax(t, am)
    elif hasattr(am, "_bcos_imgmax") and am._bcos_imgmax[2:] != ("fused",):
        del am._bcos_imgmax            # the maxima are about to grow: per-image values cached from an earlier fill are stale
    return am                          # (a range the producing launches fold in themselves grows with them: tapconv keeps or drops it)


def tapconv(a: torch.Tensor, wt: torch.Tensor, geom: dict, *, out=None, out2=None, scale_out=None,
            norm_out=None, bias=None, ch_scale=None, ch_shift=None, addend=None, mul=None, mul2=None,
            gate2=None, relu_gate=None, bcos_mode=BCOS_NONE, b=2.0, relu=False, flags=0, contraction=None,
            track_absmax=None, track_absmax2=None, max_out=1, mul_norm=None, mul_csc=None, mul_csh=None, addend_sub=0,
            col_scale=None, row_scale=None, a_sumsq=None, rowadd=None, rowadd_scale=None):
    """One launch of the generic fused implicit GEMM (include/bcos_hip.h: bcos_tapconv_ops).
    `addend_sub` = s > 1: `addend` is the dense [N, ceil(OH/s), ceil(OW/s), pitch] tensor of the output pixels on the s-grid.
    `contraction`: None = the library default, or 'f32' / 'bf16x3' / 'f16x2' for this call.
    `track_absmax` / `track_absmax2`: emit the per-pixel maxima of out / out2 (default: whenever the f16x2 contraction is
    selected; a caller that knows the reader of a tensor will not use them -- K < F16X2_MIN_K -- passes False).
    `row_scale` [rows] / `a_sumsq` [rows]: a factor of every accumulator row / the squared operand norm of a B-cos launch given
    from outside (a LayerNorm folded into the contraction, include/bcos_hip.h: bcos_epilogue.row_scale).
    `rowadd` (indexed like out) / `rowadd_scale` [output pixels]: out = acc + (rowadd_scale[pixel] rowadd + addend) in a plain gradient
    launch (bcos_epilogue.rowadd; BcosHipError with code BCOS_E_NOSUP where the launch cannot take it)."""
    lib = _l.load()
    g = TapconvGeom()
    for k in ("a_pitch", "out_pitch", "norm_pitch", "out_cgroup", "groups"):
        setattr(g, k, 0)
    for k, v in geom.items():
        setattr(g, k, int(v))
    e = Epilogue()
    tensors = dict(bias=bias, ch_scale=ch_scale, ch_shift=ch_shift, addend=addend, mul=mul, mul2=mul2,
                   gate2=gate2, relu_gate=relu_gate, out=out, out2=out2, scale_out=scale_out, norm_out=norm_out,
                   mul_norm=mul_norm, mul_csc=mul_csc, mul_csh=mul_csh, col_scale=col_scale, row_scale=row_scale, a_sumsq=a_sumsq,
                   rowadd=rowadd, rowadd_scale=rowadd_scale)
    for k, t in tensors.items():
        p = _dev(t, f"tapconv.{k}", contiguous=False)
        setattr(e, k, p.value if p is not None else None)
    e.bcos_mode = int(bcos_mode)
    e.relu = int(relu)          # 0 none, 1 ReLU, 2 GELU with constant gate
    e.b = float(b)
    e.flags = int(flags)
    e.max_out = int(max_out)
    e.addend_sub = int(addend_sub)
    if e.max_out > 1:                    # fused MaxOut: plain forward epilogue, no operand maxima of the (narrow) output
        track_absmax = track_absmax2 = False
    mode = contraction if contraction is not None else _l.get_contraction_mode()
    o = _l.Operands()
    o.a = _dev(a, "tapconv.a", contiguous=False).value
    o.wt = _dev(wt, "tapconv.wt").value
    o.contraction = {"f32": _l.CONTRACT_F32, "bf16x3": _l.CONTRACT_BF16X3, "f16x2": _l.CONTRACT_F16X2}[mode]
    # (BCOS_EPI_UNIT_NORM_W: the launch reads the RAW fp32 weight rows to gather their norms -- no pre-split image is used)
    static = getattr(wt, "_bcos_static", False) and not _NO_PRESPLIT and not (int(flags) & _l.BCOS_EPI_UNIT_NORM_W)
    keep = []
    if mode == "f16x2":
        # outputs carry their per-pixel maxima for the launch that will read them as its A operand
        pixels = int(g.N) * int(g.OH) * int(g.OW)
        for name, t, want in (("out_absmax", out, track_absmax), ("out2_absmax", out2, track_absmax2)):
            am = _out_absmax(t, pixels) if (DEFAULT_TRACK_ABSMAX if want is None else want) else None
            if am is not None:
                setattr(e, name, am.data_ptr())
                keep.append(am)
        am_a = absmax_of(a)
        ktot = int(g.TH) * int(g.TW) * int(g.C)
        if (am_a is not None and static and am_a.numel() == int(g.N) * int(g.H) * int(g.W)
                and (ktot >= F16X2_MIN_K or (int(g.C) <= 16 and ktot >= 128) or contraction == "f16x2")):
            o.a_absmax = am_a.data_ptr()
            taps = int(g.TH) * int(g.TW)               # the image is stored in the K order of a launch with this tap count
            patchable = (((taps == 9 and int(g.TH) == 3 and (int(g.Cout) > 32 or (int(g.Cout) > 8 and int(g.Q) > 64)))
                          or (taps == 16 and int(g.TH) == 4 and int(g.Cout) <= 32))
                         and int(g.in_sh) == 1 and int(g.in_sw) == 1 and int(g.dstep_h) == 1 and int(g.dstep_w) == 1 and int(g.C) % 16 == 0)
            if PATCH_LOOP and int(g.groups) <= 1 and (patchable or taps >= 25):      # (>= 25 taps, the 7 x 7 stem: image maxima replace the per-row scan of the taps)
                # 3 x 3 launches (and the 4 x 4 tap union of a depth-to-space input gradient) contract over an LDS-resident input
                # patch with one operand scale per image (include/bcos_hip.h: bcos_operands.a_imgmax): the per-image maxima, once per tensor
                im, im_fused = image_absmax(am_a, int(g.N), int(g.H) * int(g.W))
                o.a_imgmax = im[0].data_ptr()
                if im_fused:
                    o.a_imgmin_c = im[1].data_ptr()
                else:
                    o.a_imgmin = im[1].data_ptr()
                keep.append(im)
            o.wt_f16x2 = _image_of(wt, f"_bcos_wt2_t{taps}", lambda w: split_weights_f16x2(w, taps)).data_ptr()
        elif static:
            o.wt_bf16x3 = _image_of(wt, "_bcos_wt3", split_weights).data_ptr()
    elif mode == "bf16x3" and static:
        o.wt_bf16x3 = _image_of(wt, "_bcos_wt3", split_weights).data_ptr()
    # The per-image range of the maxima `out` leaves (what a 3 x 3 launch reading `out` takes as its operand scales) is folded into this
    # launch's own epilogue where the library can (specialised epilogue, >= 19 rows per image ...: it is asked) instead of a pass of
    # bcos_image_absrange ahead of the reader.  Engine passes only (their arena hands out zero-filled words without a fill launch).
    am_out = absmax_of(out) if (mode == "f16x2" and e.out_absmax) else None
    if am_out is not None:
        rec = getattr(am_out, "_bcos_imgmax", None)
        key = (int(g.N), int(g.OH) * int(g.OW))
        fuses = (FUSE_IMAGE_RANGE and _IMAGE_RANGE_READER and _ARENA is not None and int(g.P) * int(g.Q) >= 19
                 and lib.bcos_tapconv_fuses_image_range(C.byref(o), C.byref(g), C.byref(e)) == 1)
        if fuses:
            if rec is not None and rec[2:] == ("fused",) and rec[1] == key:
                img = rec[0]               # another launch of the same tensor (parity classes of a strided gradient): the range accumulates
            else:
                img = _new_absmax(2 * key[0], out.device).view(2, key[0])
                am_out._bcos_imgmax = (img, key, "fused")
            e.out_imgmax, e.out_imgmin_c = img[0].data_ptr(), img[1].data_ptr()
            keep.append(img)
        elif rec is not None:
            del am_out._bcos_imgmax        # (pixels are about to be filled that no cached range knows of)
    timing = KERNEL_TIMING
    if timing is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    code = lib.bcos_tapconv_ops(C.byref(o), C.byref(g), C.byref(e), _stream())
    if timing is not None:
        ev1.record()
        # algorithmic work of the launch: 2 M K N flops; bytes = A once + weights once + every output-sized epilogue tensor
        m_rows = int(g.N) * int(g.P) * int(g.Q)
        ktot = int(g.TH) * int(g.TW) * int(g.C)
        epi_tensors = sum(1 for t in (out, out2, scale_out, mul, mul2, gate2, relu_gate) if t is not None)
        if addend is not None:
            epi_tensors += 1.0 / max(int(addend_sub), 1) ** 2          # a subsampled addend holds 1 / s^2 of the pixels
        nbytes = int(4 * (int(g.N) * int(g.H) * int(g.W) * int(g.C) + ktot * int(g.Cout) + m_rows * int(g.Cout) * epi_tensors))
        timing.append((ev0, ev1, 2.0 * m_rows * ktot * int(g.Cout), nbytes, (m_rows, ktot, int(g.Cout))))
    _l.check(code, "bcos_tapconv")


def tapconv_group(a: torch.Tensor, wts, geoms, *, out, addend=None, mul=None):
    """Several tap sets over the same input in one call (include/bcos_hip.h: bcos_tapconv_group): the parity classes of
    a strided input gradient.  Same result as tapconv() per entry; narrow outputs run as one fused launch."""
    lib = _l.load()
    n = len(wts)
    garr = (TapconvGeom * n)()
    earr = (Epilogue * n)()
    warr = (C.c_void_p * n)()
    for i, (w, geom) in enumerate(zip(wts, geoms)):
        for k in ("a_pitch", "out_pitch", "norm_pitch", "out_cgroup", "groups"):
            setattr(garr[i], k, 0)
        for k, v in geom.items():
            setattr(garr[i], k, int(v))
        for k, t in dict(out=out, addend=addend, mul=mul).items():
            ptr = _dev(t, f"tapconv_group.{k}", contiguous=False)
            setattr(earr[i], k, ptr.value if ptr is not None else None)
        earr[i].bcos_mode, earr[i].relu, earr[i].b, earr[i].flags = BCOS_NONE, 0, 2.0, 0
        warr[i] = _dev(w, "tapconv_group.wt").value
    timing = KERNEL_TIMING
    if timing is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    code = lib.bcos_tapconv_group(_dev(a, "tapconv_group.a", contiguous=False), warr, garr, earr, n, _stream())
    if timing is not None:
        ev1.record()
        fl = sum(2.0 * q["N"] * q["P"] * q["Q"] * q["TH"] * q["TW"] * q["C"] * q["Cout"] for q in geoms)
        g0 = geoms[0]
        rows_out = sum(q["N"] * q["P"] * q["Q"] for q in geoms)
        nbytes = 4 * (g0["N"] * g0["H"] * g0["W"] * g0["C"] + rows_out * g0["Cout"] * (1 + (addend is not None) + (mul is not None)))
        timing.append((ev0, ev1, fl, nbytes, None))
    _l.check(code, "bcos_tapconv_group")


def fwd_geom(N, H, W, Cin, Cout, kh, kw, sh, sw, ph, pw, dh=1, dw=1):
    Ho, Wo = conv_out_size(H, kh, sh, ph, dh), conv_out_size(W, kw, sw, pw, dw)
    return dict(N=N, H=H, W=W, C=Cin, P=Ho, Q=Wo, in_sh=sh, in_sw=sw, dh0=-ph, dw0=-pw, dstep_h=dh,
                dstep_w=dw, TH=kh, TW=kw, OH=Ho, OW=Wo, out_sh=1, out_sw=1, out_h0=0, out_w0=0, Cout=Cout)


def conv2d_fwd(x, w, *, stride=(1, 1), padding=(0, 0), dilation=(1, 1), bias=None, b=2.0, mode=BCOS_CONV_EPS,
               ch_scale=None, ch_shift=None, addend=None, relu=False, relu_gate=None, want_scale=False,
               want_norm=False, out=None, scale_out=None, flags=0, track_absmax=None, groups=1):
    """Fused B-cos convolution.  x [N,H,W,Cin], w [Cout,kh,kw,Cin / groups] -> y [N,Ho,Wo,Cout] (+ scale, norm; grouped layers
    (bcosconv2d.py:84-140): every group has its own patch norm -- norm [N,Ho,Wo,groups] -- and the launch emits no operand
    maxima: `track_absmax` then costs one extra pass)."""
    N, H, W, Cin = x.shape
    Cout, kh, kw, Cin_w = w.shape
    G = int(groups)
    if Cin_w * G != Cin or Cout % G:
        raise BcosHipError(f"conv2d_fwd: weight has {Cin_w} input channels per group ({G} groups), activation has {Cin}")
    g = fwd_geom(N, H, W, Cin_w, Cout // G, kh, kw, stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1])
    want_track = track_absmax
    if G > 1:
        g.update(groups=G, a_pitch=Cin, out_pitch=Cout, norm_pitch=G)
        track_absmax = False
    if out is None:
        out = torch.empty((N, g["P"], g["Q"], Cout), device=x.device, dtype=torch.float32)
    if want_scale and scale_out is None:
        scale_out = torch.empty_like(out)
    norm = torch.empty((N, g["P"], g["Q"]) + ((G,) if G > 1 else ()), device=x.device, dtype=torch.float32) if want_norm else None
    if float(b) == 1.0:
        mode = BCOS_NONE
    tapconv(x, w, g, out=out, scale_out=scale_out, norm_out=norm, bias=bias, ch_scale=ch_scale,
            ch_shift=ch_shift, addend=addend, bcos_mode=mode, b=b, relu=relu, relu_gate=relu_gate, flags=flags,
            track_absmax=track_absmax)
    if G > 1 and (DEFAULT_TRACK_ABSMAX if want_track is None else want_track) and _l.get_contraction_mode() == "f16x2":
        ensure_absmax(out)
    return out, scale_out, norm


def linear_fwd(x2d, w, *, bias=None, b=2.0, want_scale=False, want_norm=False, mode=BCOS_LINEAR_EPS,
               addend=None, out=None, flags=0, col_scale=None):
    """Fused B-cos linear.  x2d [rows,Cin], w [Cout,Cin] -> y [rows,Cout]."""
    rows, Cin = x2d.shape
    Cout = w.shape[0]
    g = dict(N=1, H=1, W=rows, C=Cin, P=1, Q=rows, in_sh=1, in_sw=1, dh0=0, dw0=0, dstep_h=1, dstep_w=1,
             TH=1, TW=1, OH=1, OW=rows, out_sh=1, out_sw=1, out_h0=0, out_w0=0, Cout=Cout)
    if out is None:
        out = torch.empty((rows, Cout), device=x2d.device, dtype=torch.float32)
    scale = torch.empty_like(out) if want_scale else None
    norm = torch.empty((rows,), device=x2d.device, dtype=torch.float32) if want_norm else None
    if float(b) == 1.0:
        mode = BCOS_NONE
    tapconv(x2d, w, g, out=out, scale_out=scale, norm_out=norm, bias=bias, addend=addend, bcos_mode=mode, b=b,
            flags=flags, col_scale=col_scale)
    return out, scale, norm


def matmul_nt(a2d, bt, *, out=None, addend=None, mul=None, track_absmax=None, bias=None, row_scale=None, out2=None, mul2=None,
              track_absmax2=None, rowadd=None, rowadd_scale=None):
    """Plain fp32 GEMM on the same kernel: out[rows,N] = a2d[rows,K] @ bt[N,K]^T (no B-cos scaling).  With v = row_scale * acc + bias
    + addend:  out = v * mul, out2 = v * mul2 (the gradient epilogue of bcos_tapconv).  `rowadd` [rows,N] / `rowadd_scale` [rows]:
    + rowadd_scale[row] * rowadd (tapconv: bcos_epilogue.rowadd)."""
    rows, K = a2d.shape
    Nn = bt.shape[0]
    g = dict(N=1, H=1, W=rows, C=K, P=1, Q=rows, in_sh=1, in_sw=1, dh0=0, dw0=0, dstep_h=1, dstep_w=1,
             TH=1, TW=1, OH=1, OW=rows, out_sh=1, out_sw=1, out_h0=0, out_w0=0, Cout=Nn)
    if out is None:
        out = torch.empty((rows, Nn), device=a2d.device, dtype=torch.float32)
    tapconv(a2d, bt, g, out=out, addend=addend, mul=mul, track_absmax=track_absmax, bias=bias, row_scale=row_scale, out2=out2, mul2=mul2,
            track_absmax2=track_absmax2, rowadd=rowadd, rowadd_scale=rowadd_scale)
    return out


def matmul_nt_with_row_term(g2d, bt, x2d, rnorm, addend=None):
    """g2d @ bt^T + x2d * rnorm[:, None] (+ addend): the input gradient of a B-cos LINEAR layer with the |x| term of its scale's derivative
    (bcoslinear.py:116-142 differentiated) in the launch's epilogue -- what DgradPlan.run_with_patch_norm is for the pointwise
    convolutions; falls back to bcos_patch_norm_bwd + a plain addend where the launch cannot take it."""
    rows, Cin = x2d.shape
    if FUSE_PATCH_NORM and Cin == bt.shape[0] and Cin % 4 == 0 and x2d.is_contiguous():
        try:
            return matmul_nt(g2d, bt, addend=addend, rowadd=x2d, rowadd_scale=rnorm.reshape(-1), track_absmax=False)
        except BcosHipError as err:
            if err.code != _l.BCOS_E_NOSUP:
                raise
    term = patch_norm_bwd(x2d.view(1, 1, rows, Cin), rnorm.view(1, 1, rows), Cin, (1, 1), (1, 1), (0, 0), (1, 1),
                          addend=None if addend is None else addend.view(1, 1, rows, Cin)).view(rows, Cin)
    return matmul_nt(g2d, bt, addend=term, track_absmax=False)


def _take_taps(w, dim, idx, k):
    """w indexed by the tap list `idx` along `dim` without device index tensors where the list is the whole axis in order (1 x 1
    layers: nothing to do) or reversed (stride 1: one flip) -- a training step rebuilds every layer's plan from the updated weights."""
    if idx == list(range(k)):
        return w
    if idx == list(range(k - 1, -1, -1)):
        return torch.flip(w, (dim,))
    return w.index_select(dim, torch.tensor(idx, device=w.device))


class DgradPlan:
    """Input-gradient launches of one convolution: one tapconv per output parity class.

    For y = conv(x, w; stride s, padding p)   gx[h] = sum_r g[(h + p - r) / s] w[r]  over the r with
    (h + p - r) % s == 0.  Rows h = s*i + rho of parity class rho use the taps r = r0 + s*u,
    r0 = (rho + p) % s; with th = U-1-u the gathered g coordinate is i + dh0 + th, dh0 = (rho+p-r0)/s - U + 1.
    """

    def __init__(self, w_oihw: torch.Tensor, stride, padding, dilation=(1, 1), groups: int = 1, transient: bool = False):
        """`transient`: see mark_static.  `groups` > 1 (w_oihw [Cout, Cin / groups, kh, kw], the layout of nn.Conv2d): every class is ONE grouped launch
        (bcos_tapconv_geom.groups) -- group g contracts its Cout / groups gradient channels with its own transposed filters and
        writes its Cin / groups columns; Cin below is the layer's total input width."""
        Cout, Cin_g, kh, kw = w_oihw.shape
        G = int(groups)
        if G < 1 or Cout % G:
            raise BcosHipError("dgrad: out_channels must be divisible by groups")
        if G > 1 and ((Cout // G) % 4 or Cin_g % 4):
            raise BcosHipError("dgrad: grouped layers need in_channels / groups and out_channels / groups to be multiples of 4")
        Cin = Cin_g * G
        sh, sw = stride
        ph, pw = padding
        if (dilation[0] != 1 or dilation[1] != 1) and (sh != 1 or sw != 1):
            raise BcosHipError("dgrad with dilation > 1 and stride > 1 is not supported")
        self.stride, self.padding, self.dilation = (sh, sw), (ph, pw), tuple(dilation)
        self.Cin, self.Cout, self.k, self.groups = Cin, Cout, (kh, kw), G
        self.classes = []     # (rho_h, rho_w, TH, TW, dh0, dw0, weight [Cin,TH,TW,Cout] or None)
        for (rh, rw, rs_h, rs_w, dh0, dw0, step_h, step_w) in self.class_taps((kh, kw), (sh, sw), (ph, pw), dilation):
            if len(rs_h) == 0 or len(rs_w) == 0:
                self.classes.append((rh, rw, 0, 0, 0, 0, 1, 1, None))
                continue
            sub = _take_taps(_take_taps(w_oihw, 2, rs_h, kh), 3, rs_w, kw)      # [Cout,Cin/G,TH,TW]
            if G == 1:
                wt = sub.permute(1, 2, 3, 0).contiguous()               # [Cin,TH,TW,Cout]
            else:                                                       # [G Cin/G, TH, TW, Cout/G]: the groups' transposed filters, stacked
                cg = Cout // G
                wt = torch.cat([sub[k * cg:(k + 1) * cg].permute(1, 2, 3, 0) for k in range(G)], 0).contiguous()
            mark_static(wt, transient)     # a DgradPlan is built once per weight version (engine plan / WeightCache)
            self.classes.append((rh, rw, len(rs_h), len(rs_w), dh0, dw0, step_h, step_w, wt))
        self.has_empty = any(c[8] is None for c in self.classes)
        self.transient = bool(transient)
        self._d2s = {}        # channel pitch -> (weights [sh*sw*pitch, TH, TW, Cout], TH, TW, dh0, dw0), see _depth_to_space

    @classmethod
    def class_taps(cls, k, stride, padding, dilation=(1, 1)):
        """The parity classes of the input gradient of a k / stride / padding convolution, in the order of `classes`:
        [(rho_h, rho_w, rs_h, rs_w, dh0, dw0, step_h, step_w)] -- rs_h / rs_w: the filter taps of the class in the order its launch walks them
        (empty: no tap reaches the class)."""
        out = []
        for rh in range(stride[0]):
            rs_h, dh0, step_h = cls._taps(rh, stride[0], padding[0], k[0], dilation[0])
            for rw in range(stride[1]):
                rs_w, dw0, step_w = cls._taps(rw, stride[1], padding[1], k[1], dilation[1])
                out.append((rh, rw, rs_h, rs_w, dh0, dw0, step_h, step_w))
        return out

    @classmethod
    def from_banks(cls, Cin, Cout, k, stride, padding, dilation, banks):
        """A plan over weight banks made elsewhere (WeightPrepBatch: the training plans rebuild every bank of a step by ONE launch): `banks`
        [i] = the [Cin, TH, TW, Cout] bank of class i of class_taps(...), None where the class has no tap.  groups = 1."""
        if (dilation[0] != 1 or dilation[1] != 1) and (stride[0] != 1 or stride[1] != 1):
            raise BcosHipError("dgrad with dilation > 1 and stride > 1 is not supported")
        self = cls.__new__(cls)
        self.stride, self.padding, self.dilation = tuple(stride), tuple(padding), tuple(dilation)
        self.Cin, self.Cout, self.k, self.groups = int(Cin), int(Cout), tuple(k), 1
        self.classes = []
        for (rh, rw, rs_h, rs_w, dh0, dw0, step_h, step_w), wt in zip(cls.class_taps(k, stride, padding, dilation), banks):
            if len(rs_h) == 0 or len(rs_w) == 0:
                self.classes.append((rh, rw, 0, 0, 0, 0, 1, 1, None))
            else:
                self.classes.append((rh, rw, len(rs_h), len(rs_w), dh0, dw0, step_h, step_w, wt))
        self.has_empty = any(c[8] is None for c in self.classes)
        self.transient = False
        self._d2s = {}
        return self

    def _depth_to_space(self, pitch: int):
        """All parity classes of a narrow strided input gradient as ONE contraction (bcos_tapconv_geom.out_cgroup): rows =
        coarse positions (i, j), columns = (parity class, channel), taps = the union of the classes' tap windows over the
        gradient, with zero weights where a class does not use a tap.  The 2 x 2 classes of the 7x7 / 2 stem gradient
        (16 / 12 / 12 / 9 taps over 64 channels) become one K = 1024, N = 32 launch on the MFMA tiles instead of four
        6-column launches."""
        hit = self._d2s.get(pitch)
        if hit is not None:
            return hit
        sh, sw = self.stride
        live = [c for c in self.classes if c[8] is not None]
        dh0 = min(c[4] for c in live)
        dw0 = min(c[5] for c in live)
        TH = max(c[4] + c[2] for c in live) - dh0
        TW = max(c[5] + c[3] for c in live) - dw0
        ref = live[0][8]
        wc = torch.zeros((sh * sw * pitch, TH, TW, self.Cout), device=ref.device, dtype=ref.dtype)
        for (rh, rw, th, tw, h0, w0, _, _, wt) in live:
            base = (rh * sw + rw) * pitch
            wc[base:base + self.Cin, h0 - dh0:h0 - dh0 + th, w0 - dw0:w0 - dw0 + tw] = wt
        mark_static(wc, self.transient)
        if not self.transient:
            publish_cached(wc)
        self._d2s[pitch] = (wc, TH, TW, dh0, dw0)
        return self._d2s[pitch]

    @staticmethod
    def _taps(rho, s, p, k, d):
        if s == 1:
            # gx[h] = sum_r g[h + p - r*d] w[r]; th = k-1-r -> offset = p - (k-1)*d + th*d
            return list(range(k - 1, -1, -1)), p - (k - 1) * d, d
        r0 = (rho + p) % s
        rs = list(range(r0, k, s))
        U = len(rs)
        if U == 0:
            return [], 0, 1
        e = (rho + p - r0) // s
        return rs[::-1], e - U + 1, 1

    @property
    def subsampled(self) -> int:
        """s when this is the gradient of a 1x1 / stride-s / unpadded convolution (a ResNet shortcut): gx is zero off the
        s-grid and `run_compact` returns the grid pixels alone; 0 otherwise."""
        sh, sw = self.stride
        return sh if (self.k == (1, 1) and sh == sw and sh > 1 and self.padding == (0, 0) and self.groups == 1) else 0

    def run_compact(self, glin, **track):
        """glin [N,Ho,Wo,Cout] -> the non-zero pixels of gx as a dense [N,Ho,Wo,Cin] tensor (gx[:, ::s, ::s] of `run`), for
        a reader that takes it as a subsampled addend (bcos_epilogue.addend_sub) instead of a zero-filled full-size tensor."""
        assert self.subsampled
        N, Ho, Wo, Cout = glin.shape
        wt = self.classes[0][8]
        out = torch.empty((N, Ho, Wo, self.Cin), device=glin.device, dtype=torch.float32)
        g = dict(N=N, H=Ho, W=Wo, C=Cout, P=Ho, Q=Wo, in_sh=1, in_sw=1, dh0=0, dw0=0, dstep_h=1, dstep_w=1, TH=1, TW=1,
                 OH=Ho, OW=Wo, out_sh=1, out_sw=1, out_h0=0, out_w0=0, Cout=self.Cin)
        tapconv(glin, wt, g, out=out, **track)
        return out

    def run(self, glin, H, W, *, out=None, **epi):
        """glin [N,Ho,Wo,Cout] -> gx [N,H,W,Cin]; **epi are tapconv epilogue tensors indexed like gx.
        `out` may be wider than Cin (padded channel pitch); only the first Cin channels are written."""
        N, Ho, Wo, Cout = glin.shape
        track = {k: epi.pop(k) for k in ("track_absmax", "track_absmax2") if k in epi}
        epi = {k: v for k, v in epi.items() if v is not None}
        zero_filled = False
        if out is None:
            if self.has_empty and not epi:
                out = torch.zeros((N, H, W, self.Cin), device=glin.device, dtype=torch.float32)
                zero_filled = True
            else:
                out = torch.empty((N, H, W, self.Cin), device=glin.device, dtype=torch.float32)
        pitch = out.shape[-1]
        sh, sw = self.stride
        narrow = (self.Cin <= 8 and len(self.classes) > 1 and not self.has_empty and set(epi) <= {"addend", "mul"}
                  and self.dilation == (1, 1) and self.groups == 1)
        G = self.groups
        grouped = {}
        if G > 1:       # one launch per class for all groups; grouped launches emit no operand maxima (one extra pass where asked for)
            if Cout != self.Cout or pitch < self.Cin:
                raise BcosHipError("dgrad: gradient / output width does not match the grouped layer")
            grouped = dict(groups=G, a_pitch=Cout)
            want_track = bool(track.get("track_absmax")) or bool(track.get("track_absmax2"))
            track = dict(track_absmax=False, track_absmax2=False)
        if narrow and not _NO_D2S and pitch % 4 == 0 and pitch <= 16 and H % sh == 0 and W % sw == 0:
            # one launch for all parity classes: columns = (class, channel), depth-to-space output mapping
            wc, TH, TW, dh0, dw0 = self._depth_to_space(pitch)
            g = dict(N=N, H=Ho, W=Wo, C=Cout, P=H // sh, Q=W // sw, in_sh=1, in_sw=1, dh0=dh0, dw0=dw0, dstep_h=1, dstep_w=1,
                     TH=TH, TW=TW, OH=H, OW=W, out_sh=sh, out_sw=sw, out_h0=0, out_w0=0, Cout=sh * sw * pitch, out_pitch=pitch,
                     out_cgroup=pitch)
            tapconv(glin, wc, g, out=out, track_absmax=False, track_absmax2=False, **epi)
            return out
        if narrow and not _NO_GROUP:
            # narrow output (the stem gradient): all parity classes in one launch, the input patch staged once
            geoms, wts = [], []
            for (rh, rw, TH, TW, dh0, dw0, step_h, step_w, wt) in self.classes:
                P = (H - rh + sh - 1) // sh
                Q = (W - rw + sw - 1) // sw
                if P <= 0 or Q <= 0:
                    continue
                geoms.append(dict(N=N, H=Ho, W=Wo, C=Cout, P=P, Q=Q, in_sh=1, in_sw=1, dh0=dh0, dw0=dw0, dstep_h=step_h,
                                  dstep_w=step_w, TH=TH, TW=TW, OH=H, OW=W, out_sh=sh, out_sw=sw, out_h0=rh, out_w0=rw,
                                  Cout=self.Cin, out_pitch=pitch))
                wts.append(wt)
            tapconv_group(glin, wts, geoms, out=out, **epi)
            return out
        for (rh, rw, TH, TW, dh0, dw0, step_h, step_w, wt) in self.classes:
            P = (H - rh + sh - 1) // sh
            Q = (W - rw + sw - 1) // sw
            if P <= 0 or Q <= 0:
                continue
            if wt is None:
                # no tap reaches this parity class: the gradient there is whatever the epilogue adds to 0
                if not zero_filled:
                    self._empty_class(out, N, H, W, rh, rw, P, Q, epi)
                continue
            g = dict(N=N, H=Ho, W=Wo, C=Cout // G, P=P, Q=Q, in_sh=1, in_sw=1, dh0=dh0, dw0=dw0, dstep_h=step_h,
                     dstep_w=step_w, TH=TH, TW=TW, OH=H, OW=W, out_sh=sh, out_sw=sw, out_h0=rh, out_w0=rw,
                     Cout=self.Cin // G, out_pitch=pitch, **grouped)
            tapconv(glin, wt, g, out=out, **epi, **track)
        if G > 1 and want_track:
            ensure_absmax(out)
            if epi.get("out2") is not None:
                ensure_absmax(epi["out2"])
        return out

    @property
    def pointwise(self) -> bool:
        """the gradient of a 1 x 1 / stride-1 / unpadded, ungrouped convolution: every output pixel of run() is one row of ONE launch"""
        return self.k == (1, 1) and self.stride == (1, 1) and self.padding == (0, 0) and self.groups == 1 and len(self.classes) == 1

    def run_with_patch_norm(self, glin, x, rnorm, cin, H, W, addend=None):
        """run(glin, H, W) + the patch-norm term of the B-cos scale's derivative, x * (sum of rnorm over the patches that contain the pixel)
        (+ `addend`: a gradient that reaches the layer input by another path).  Pointwise layers (`pointwise`: patch = pixel) take the
        term inside the launch's epilogue (bcos_epilogue.rowadd: one read of x instead of bcos_patch_norm_bwd_add's pass that writes a
        tensor for the launch to read back); every other geometry, and a launch the library answers BCOS_E_NOSUP for, as before."""
        N = glin.shape[0]
        if FUSE_PATCH_NORM and self.pointwise and x.shape[-1] == self.Cin and cin == self.Cin and self.Cin % 4 == 0 and x.is_contiguous():
            try:
                return self.run(glin, H, W, addend=addend, rowadd=x, rowadd_scale=rnorm.reshape(-1))
            except BcosHipError as err:
                if err.code != _l.BCOS_E_NOSUP:
                    raise
        Ho, Wo = glin.shape[1], glin.shape[2]
        return self.run(glin, H, W, addend=patch_norm_bwd(x, rnorm.view(N, Ho, Wo), cin, self.k, self.stride, self.padding, self.dilation,
                                                          addend=addend))

    def _empty_class(self, out, N, H, W, rh, rw, P, Q, epi):
        sh, sw = self.stride
        view = out[:, rh::sh, rw::sw, :self.Cin]
        drop_absmax(out)          # these pixels are filled outside the kernels: no per-pixel maxima for `out`
        addend, mul = epi.get("addend"), epi.get("mul")
        if addend is None:
            view.zero_()
        else:
            v = addend[:, rh::sh, rw::sw, :]
            view.copy_(v if mul is None else v * mul[:, rh::sh, rw::sw, :])
        for k in ("out2", "scale_out", "mul2", "gate2"):
            if epi.get(k) is not None:
                raise BcosHipError(f"dgrad parity class without taps does not support epilogue field {k}")
        if epi.get("relu"):
            raise BcosHipError("dgrad parity class without taps does not support relu")


class WeightPrepBatch:
    """Weight banks and f16x2 images of MANY layers, kept in buffers of their own and rebuilt from the layers' parameters by ONE call
    (include/bcos_hip.h: bcos_weight_prep_batch).  A training step changes every weight once; per layer, the forward bank (NHWC filters) and
    its image, the transposed / tap-reversed banks of the input-gradient classes and theirs, each made by two to five small launches
    (layout copy, flip, row scale, split), were ~350 launches of a ResNet-50 step.  add_*() registers a job and hands out the bank tensor
    -- marked static, with its image attached where tapconv looks for it -- and run() refreshes all of them on the current stream."""

    def __init__(self, device):
        self.device = torch.device(device)
        self._jobs, self._srcs, self._banks, self._keep = [], [], [], []
        self._table = self._row_max = None
        self._max_rows = self._max_ktot = self._rows_total = 0

    def _add(self, src, bank_shape, rows, channels, Cp, taps, row_stride, ch_stride, tap_offsets):
        if taps > _l.PREP_MAX_TAPS or src.dtype != torch.float32 or not src.is_contiguous():
            return None
        lib = _l.load()
        nbytes = C.c_int64(0)
        _l.check(lib.bcos_split_weights_f16x2_bytes(rows, taps * Cp, C.byref(nbytes)), "bcos_split_weights_f16x2_bytes")
        bank = torch.empty(bank_shape, device=self.device, dtype=torch.float32)
        image = torch.empty(nbytes.value, device=self.device, dtype=torch.uint8)
        j = _l.WeightPrepJob()
        j.src, j.bank, j.image = src.data_ptr(), bank.data_ptr(), image.data_ptr()
        j.rows, j.channels, j.Cp, j.taps, j.row_stride, j.ch_stride = rows, channels, Cp, taps, row_stride, ch_stride
        for t, o in enumerate(tap_offsets):
            j.tap_offset[t] = int(o)
        j.row_offset = self._rows_total
        self._rows_total += (rows + 127) & ~127
        self._max_ktot = max(self._max_ktot, taps * Cp)
        mark_static(bank)
        setattr(bank, f"_bcos_wt2_t{taps}", (bank._version, image))      # (where _image_of looks; the bank is never written through torch: its version stays)
        self._jobs.append(j)
        self._srcs.append(src)
        self._banks.append(bank)
        self._keep.append(image)
        self._max_rows = max(self._max_rows, rows)
        self._table = None
        return bank

    def add_forward(self, w_oihw):
        """-> the [Cout, kh, kw, Cp] bank of OIHW weights (Cp = Cin rounded up to 4), or None where the batch cannot make it"""
        Cout, Cin, kh, kw = w_oihw.shape
        Cp = (Cin + 3) & ~3
        return self._add(w_oihw, (Cout, kh, kw, Cp), Cout, Cin, Cp, kh * kw, Cin * kh * kw, kh * kw, range(kh * kw))

    def add_linear(self, w2d):
        """-> (the [Cout, Cp] bank of a linear layer's [Cout, Cin] weight, the [Cin, Coutp] bank of its transpose: what the forward launch
        and the input-gradient launch of the layer read), or None"""
        Cout, Cin = w2d.shape
        Cp, Coutp = (Cin + 3) & ~3, (Cout + 3) & ~3
        fwd = self._add(w2d, (Cout, Cp), Cout, Cin, Cp, 1, Cin, 1, (0,))
        tr = self._add(w2d, (Cin, Coutp), Cin, Cout, Coutp, 1, 1, Cin, (0,))
        return None if (fwd is None or tr is None) else (fwd, tr)

    def add_dgrad(self, w_oihw, stride, padding, dilation=(1, 1)):
        """-> DgradPlan over banks of this batch (its K dimension = Cout rounded up to 4 with zero filters), or None"""
        Cout, Cin, kh, kw = w_oihw.shape
        Cp = (Cout + 3) & ~3
        banks = []
        for (rh, rw, rs_h, rs_w, *_rest) in DgradPlan.class_taps((kh, kw), stride, padding, dilation):
            if len(rs_h) == 0 or len(rs_w) == 0:
                banks.append(None)
                continue
            offs = [a * kw + b for a in rs_h for b in rs_w]
            bank = self._add(w_oihw, (Cin, len(rs_h), len(rs_w), Cp), Cin, Cout, Cp, len(offs), kh * kw, Cin * kh * kw, offs)
            if bank is None:
                return None
            banks.append(bank)
        return DgradPlan.from_banks(Cin, Cp, (kh, kw), stride, padding, dilation, banks)

    def run(self):
        """refresh every bank and image from the current values of the parameters (one launch on the current stream)"""
        if not self._jobs:
            return
        moved = any(j.src != s.data_ptr() for j, s in zip(self._jobs, self._srcs))
        if self._table is None or moved:
            for j, s in zip(self._jobs, self._srcs):
                j.src = s.data_ptr()
            raw = b"".join(bytes(j) for j in self._jobs)
            self._table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        for b in self._banks:               # (an image tapconv made lazily of an earlier version of a bank -- the bf16x3 one -- is stale now)
            if hasattr(b, "_bcos_wt3"):
                del b._bcos_wt3
        if self._row_max is None or self._row_max.numel() != self._rows_total:
            self._row_max = torch.empty(self._rows_total, device=self.device, dtype=torch.int32)
        self._row_max.zero_()               # (the rows' maxima meet there by atomicMax)
        _l.check(_l.load().bcos_weight_prep_batch(C.c_void_p(self._table.data_ptr()), len(self._jobs), self._max_rows, self._max_ktot,
                                                  C.c_void_p(self._row_max.data_ptr()), _stream()), "bcos_weight_prep_batch")


def weight_rownorm_scale(w2d, gain=None):
    lib = _l.load()
    out = torch.empty_like(w2d)
    rows, cols = w2d.shape[0], w2d[0].numel()
    _l.check(lib.bcos_weight_rownorm_scale(_dev(w2d, "w"), _dev(gain, "gain"), _dev(out, "out"), rows, cols, _stream()),
             "bcos_weight_rownorm_scale")
    return out


def rows_normalize(x2d, want_y=True, want_inv=False):
    """(x / ||x||_2 per row, 1 / ||x||_2 per row) of a [rows, C] tensor (include/bcos_hip.h: bcos_rows_normalize)."""
    lib = _l.load()
    rows, Cc = x2d.shape
    y = torch.empty_like(x2d) if want_y else None
    inv = torch.empty((rows,), device=x2d.device, dtype=torch.float32) if want_inv else None
    _l.check(lib.bcos_rows_normalize(_dev(x2d, "x"), _dev(y, "y"), _dev(inv, "inv"), rows, Cc, _stream()), "bcos_rows_normalize")
    return y, inv


def cosine_grad(u2d, w2d, l, inv, coef=None):
    """coef * inv * (w - l u) per row: gradient of the cosine logit l = u . w w.r.t. the un-normalised feature (bcos_cosine_grad)."""
    lib = _l.load()
    rows, Cc = u2d.shape
    out = torch.empty_like(u2d)
    _l.check(lib.bcos_cosine_grad(_dev(u2d, "u"), _dev(w2d, "w"), _dev(l, "l"), _dev(inv, "inv"), _dev(coef, "coef"), _dev(out, "out"),
                                  rows, Cc, _stream()), "bcos_cosine_grad")
    return out


def mul(a, b, out=None):
    lib = _l.load()
    if out is None:
        out = torch.empty_like(a)
    _l.check(lib.bcos_mul(_dev(a, "a"), _dev(b, "b"), _dev(out, "out"), a.numel(), _stream()), "bcos_mul")
    return out


def maxout_expand(gy2d, t2d, max_out):
    """glin[r, c] = gy[r, c // M] * t[r, c]: backward of the fused MaxOut (include/bcos_hip.h: bcos_maxout_expand)."""
    lib = _l.load()
    rows, Cout = t2d.shape
    out = torch.empty_like(t2d)
    _l.check(lib.bcos_maxout_expand(_dev(gy2d, "gy"), _dev(t2d, "t"), _dev(out, "glin"), rows, Cout, int(max_out), _stream()),
             "bcos_maxout_expand")
    return out


def maxout_scale(lin2d, norm, Cout, max_out, b, groups=1, want_scale=False, want_argmax=False, out=None):
    lib = _l.load()
    rows = lin2d.shape[0]
    y = out if out is not None else torch.empty((rows, Cout), device=lin2d.device, dtype=torch.float32)
    scale = torch.empty((rows, Cout), device=lin2d.device, dtype=torch.float32) if want_scale else None
    arg = torch.empty((rows, Cout), device=lin2d.device, dtype=torch.int32) if want_argmax else None
    argp = C.c_void_p(arg.data_ptr()) if arg is not None else None
    _l.check(lib.bcos_maxout_scale(_dev(lin2d, "lin"), _dev(norm, "norm"), _dev(y, "y"), _dev(scale, "scale"), argp,
                                   rows, Cout, max_out, groups, float(b), _stream()), "bcos_maxout_scale")
    return y, scale, arg


def _fused_absmax(t: torch.Tensor, want: bool):
    """side tensor for a kernel that can emit the per-pixel maxima of its output `t` itself (f16x2 contraction only)"""
    if not want or _l.get_contraction_mode() != "f16x2":
        return None
    am = _new_absmax(t.numel() // t.shape[-1], t.device)
    _attach_absmax(t, am)
    return am


def prep_input(x_nchw, mean6, std6, cpad=8, add_inverse=False, want_absmax=False):
    """`want_absmax`: the kernel also emits the per-pixel maxima of its output (what ensure_absmax would compute in a second
    pass) for the f16x2 contraction that reads it."""
    lib = _l.load()
    N, Cx, H, W = x_nchw.shape
    out = torch.empty((N, H, W, cpad), device=x_nchw.device, dtype=torch.float32)
    am = _fused_absmax(out, want_absmax)
    _l.check(lib.bcos_prep_input(_dev(x_nchw, "x"), _dev(out, "out"), _dev(mean6, "mean"), _dev(std6, "std"),
                                 C.c_void_p(am.data_ptr()) if am is not None else None, N, Cx, H, W,
                                 cpad, int(add_inverse), _stream()), "bcos_prep_input")
    return out


def finalize_explanation(gxn, x_nchw, std6, add_inverse=False, want_weights=True, want_contrib=True, weights_out=None, contrib_out=None):
    """`weights_out` / `contrib_out`: write into these [N,6,H,W] / [N,H,W] tensors (slices of a larger batch) instead of new ones."""
    lib = _l.load()
    N, H, W, cpad = gxn.shape
    Cx = x_nchw.shape[1]
    wout = (weights_out if weights_out is not None else torch.empty((N, 6, H, W), device=gxn.device, dtype=torch.float32)) if want_weights else None
    cout = (contrib_out if contrib_out is not None else torch.empty((N, H, W), device=gxn.device, dtype=torch.float32)) if want_contrib else None
    _l.check(lib.bcos_finalize_explanation(_dev(gxn, "gxn"), _dev(x_nchw, "x"), _dev(std6, "std"), _dev(wout, "w"),
                                           _dev(cout, "c"), N, Cx, H, W, cpad, int(add_inverse), _stream()),
             "bcos_finalize_explanation")
    return wout, cout


def contrib_map(x_nchw, gx_nchw):
    lib = _l.load()
    N, Cc, H, W = x_nchw.shape
    out = torch.empty((N, H, W), device=x_nchw.device, dtype=torch.float32)
    _l.check(lib.bcos_contrib_map(_dev(x_nchw, "x"), _dev(gx_nchw, "gx"), _dev(out, "out"), N, Cc, H, W, _stream()),
             "bcos_contrib_map")
    return out


def avgpool2d_fwd(x, k, s, p, out=None, want_absmax=False):
    """`want_absmax`: also emit the per-pixel maxima of the pooled tensor (k in {2, 3}, C / 4 a power of two <= 64: the pools of the
    ResNets; other shapes fall back to the separate pass of ensure_absmax)."""
    lib = _l.load()
    N, H, W, Cc = x.shape
    OH, OW = conv_out_size(H, k, s, p), conv_out_size(W, k, s, p)
    y = out if out is not None else torch.empty((N, OH, OW, Cc), device=x.device, dtype=torch.float32)
    c4 = Cc // 4
    fused = want_absmax and k in (2, 3) and Cc % 4 == 0 and c4 <= 64 and (c4 & (c4 - 1)) == 0 and H * W * c4 < (1 << 31)
    am = _fused_absmax(y, fused)
    _l.check(lib.bcos_avgpool2d_fwd_absmax(_dev(x, "x"), _dev(y, "y"), C.c_void_p(am.data_ptr()) if am is not None else None,
                                           N, H, W, Cc, k, s, p, OH, OW, _stream()), "bcos_avgpool2d_fwd")
    if want_absmax and am is None:
        ensure_absmax(y)
    return y


def avgpool2d_bwd(gy, H, W, k, s, p, mul=None, out=None, want_absmax=False):
    """`want_absmax`: also emit the per-pixel maxima of the result (channel counts whose C / 4 is a power of two <= 64; other
    widths fall back to the separate pass of ensure_absmax)."""
    lib = _l.load()
    N, OH, OW, Cc = gy.shape
    if out is None:
        out = torch.empty((N, H, W, Cc), device=gy.device, dtype=torch.float32)
    c4 = Cc // 4
    fused = want_absmax and Cc % 4 == 0 and c4 <= 64 and (c4 & (c4 - 1)) == 0
    am = _fused_absmax(out, fused)
    _l.check(lib.bcos_avgpool2d_bwd(_dev(gy, "gy"), _dev(mul, "mul"), _dev(out, "gx"), C.c_void_p(am.data_ptr()) if am is not None else None,
                                    N, H, W, Cc, k, s, p, OH, OW, _stream()), "bcos_avgpool2d_bwd")
    if want_absmax and am is None:
        ensure_absmax(out)
    return out


def global_avgpool_logits(x, temperature=None, bias=None):
    lib = _l.load()
    N, H, W, Cc = x.shape
    y = torch.empty((N, Cc), device=x.device, dtype=torch.float32)
    inv_t = 1.0 if temperature is None else 1.0 / float(temperature)
    _l.check(lib.bcos_global_avgpool_logits(_dev(x, "x"), _dev(y, "y"), N, H * W, Cc, inv_t, 0.0 if bias is None else float(bias),
                                            _stream()), "bcos_global_avgpool_logits")
    return y


def head_onehot_grad(cls, scale, temperature=None, out=None):
    lib = _l.load()
    N, H, W, Cc = scale.shape
    if cls.dtype != torch.int64 or not cls.is_cuda:
        raise BcosHipError("head_onehot_grad: cls must be an int64 HIP tensor")
    if out is None:
        out = torch.empty_like(scale)
    inv_t = 1.0 if temperature is None else 1.0 / float(temperature)
    _l.check(lib.bcos_head_onehot_grad(C.c_void_p(cls.data_ptr()), _dev(scale, "scale"), _dev(out, "glin"), N, H * W, Cc, inv_t,
                                       _stream()), "bcos_head_onehot_grad")
    return out


def stream_copy(src: torch.Tensor, dst: torch.Tensor = None) -> torch.Tensor:
    """dst <- src (contiguous fp32, numel % 4 == 0) by the library's own streaming kernel on the current stream
    (include/bcos_hip.h: bcos_stream_copy): the bandwidth reference of bench.py."""
    dst = torch.empty_like(src) if dst is None else dst
    if src.dtype != torch.float32 or dst.dtype != torch.float32 or not src.is_contiguous() or not dst.is_contiguous() or src.numel() != dst.numel():
        raise BcosHipError("stream_copy: contiguous fp32 tensors of equal size")
    _l.check(_l.load().bcos_stream_copy(_dev(src, "src"), _dev(dst, "dst"), src.numel(), _stream()), "bcos_stream_copy")
    return dst


def check_targets(targets, n_logits: int, what: str = "targets"):
    """Class indices handed to an explanation pass, validated ONCE on the host before any launch: the reference explains
    `out[0, idx]` (bcos/common.py:170-176), so an index in [-K, -1] counts from the end and anything outside [-K, K) is an IndexError --
    never an out-of-range read on the device (the rank-one head gradient reads column cls[n] of the head's stored scale and row cls[n] of
    its weights).  Returns an int64 tensor with the negative indices wrapped; None stays None.  A tensor that already lives on the
    device costs one synchronisation of its stream here (its extrema are read back); host tensors and lists cost nothing."""
    if targets is None:
        return None
    t = torch.as_tensor(targets)
    if t.dtype.is_floating_point or t.dtype == torch.bool or t.is_complex():
        raise TypeError(f"{what}: class indices must be integers, got {t.dtype}")
    t = t.to(torch.int64)
    if t.numel():
        lo, hi = (int(v) for v in torch.aminmax(t))
        if lo < -n_logits or hi >= n_logits:
            raise IndexError(f"{what}: index {hi if hi >= n_logits else lo} is out of bounds for {n_logits} logits")
        if lo < 0:
            t = torch.where(t < 0, t + n_logits, t)
    return t


def head_rank1_grad(cls, scale, w, temperature=None, row_scale=None, mul=None, want_out2=False, want_absmax=False, mul2=None, gate2=None,
                    gate2_from_mul=False, want_absmax2=False):
    """The one-hot head gradient carried through the head's linear map in one launch (include/bcos_hip.h: bcos_head_rank1_grad[_ex]):
    scale [N, R, K], w [K, D] -> (out [N R, D] = v * mul, out2 or None) with v = inv_t / R * scale[n, r, cls_n] * row_scale * w[cls_n];
    out2 = v [* mul2], zeroed where gate2 <= 0 / where the low mantissa bit of mul is clear (the gradient epilogue's second output)."""
    lib = _l.load()
    N, R, K = scale.shape
    D = w.shape[1]
    if cls.dtype != torch.int64 or not cls.is_cuda:
        raise BcosHipError("head_rank1_grad: cls must be an int64 HIP tensor")
    for name, t in (("mul", mul), ("mul2", mul2), ("gate2", gate2)):
        if t is not None and (t.numel() != N * R * D or not t.is_contiguous()):
            raise BcosHipError(f"head_rank1_grad: {name} {tuple(t.shape)} does not match {(N * R, D)}")
    if tuple(w.shape) != (K, D):
        raise BcosHipError(f"head_rank1_grad: w {tuple(w.shape)} does not match scale {tuple(scale.shape)}")
    want_out2 = bool(want_out2 or mul2 is not None or gate2 is not None or gate2_from_mul or want_absmax2)
    out = torch.empty((N * R, D), device=scale.device, dtype=torch.float32)
    out2 = torch.empty_like(out) if want_out2 else None
    f16 = _l.get_contraction_mode() == "f16x2"
    am = torch.empty((N * R,), device=scale.device, dtype=torch.int32) if (want_absmax and f16) else None
    am2 = torch.empty((N * R,), device=scale.device, dtype=torch.int32) if (want_absmax2 and f16 and want_out2) else None
    inv_t = 1.0 if temperature is None else 1.0 / float(temperature)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None      # noqa: E731
    _l.check(lib.bcos_head_rank1_grad_ex(ptr(cls), _dev(scale, "scale"), _dev(w, "w"), _dev(row_scale, "row_scale"), _dev(mul, "mul"),
                                         _dev(mul2, "mul2"), _dev(gate2, "gate2"), int(bool(gate2_from_mul)), _dev(out, "out"), _dev(out2, "out2"),
                                         ptr(am), ptr(am2), N, R, K, D, inv_t, _stream()), "bcos_head_rank1_grad")
    if am is not None:
        _attach_absmax(out, am)
    if am2 is not None:
        _attach_absmax(out2, am2)
    return out, out2


def argmax_rows(x2d):
    lib = _l.load()
    N, Cc = x2d.shape
    idx = torch.empty((N,), device=x2d.device, dtype=torch.int64)
    val = torch.empty((N,), device=x2d.device, dtype=torch.float32)
    _l.check(lib.bcos_argmax_rows(_dev(x2d, "x"), C.c_void_p(idx.data_ptr()), _dev(val, "val"), N, Cc, _stream()), "bcos_argmax_rows")
    return idx, val


def channel_affine(x, scale, shift=None, relu=False, out=None):
    lib = _l.load()
    Cc = x.shape[-1]
    if out is None:
        out = torch.empty_like(x)
    _l.check(lib.bcos_channel_affine(_dev(x, "x"), _dev(scale, "scale"), _dev(shift, "shift"), _dev(out, "y"), x.numel() // Cc, Cc,
                                     int(bool(relu)), _stream()), "bcos_channel_affine")
    return out


def channel_affine_add(x, scale, shift, addend, relu=False, out=None):
    """out = [relu](x * scale[c] + shift[c] + addend) (include/bcos_hip.h: bcos_channel_affine_add)."""
    lib = _l.load()
    Cc = x.shape[-1]
    if out is None:
        out = torch.empty_like(x)
    _l.check(lib.bcos_channel_affine_add(_dev(x, "x"), _dev(scale, "scale"), _dev(shift, "shift"), _dev(addend, "addend"), _dev(out, "y"),
                                         x.numel() // Cc, Cc, int(bool(relu)), _stream()), "bcos_channel_affine_add")
    return out


def channel_affine_rows(x, scale, shift=None, addend=None, relu=False):
    """out = [relu](x * scale[c] + shift[c] (+ addend)) with the per-row maxima of `out` attached (include/bcos_hip.h:
    bcos_channel_affine_rows): the contraction that reads `out` then runs the split-f16 loop."""
    lib = _l.load()
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    out = torch.empty_like(x)
    am = torch.empty((rows,), device=x.device, dtype=torch.int32)
    _l.check(lib.bcos_channel_affine_rows(_dev(x, "x"), _dev(scale, "scale"), _dev(shift, "shift"), _dev(addend, "addend"), _dev(out, "y"),
                                          C.c_void_p(am.data_ptr()), rows, Cc, int(bool(relu)), _stream()), "bcos_channel_affine_rows")
    _attach_absmax(out, am)
    return out


def relu_bwd(g, act, out=None):
    """out = act > 0 ? g : 0 (include/bcos_hip.h: bcos_relu_bwd)."""
    lib = _l.load()
    if out is None:
        out = torch.empty_like(g)
    _l.check(lib.bcos_relu_bwd(_dev(g, "g"), _dev(act, "act"), _dev(out, "out"), g.numel(), _stream()), "bcos_relu_bwd")
    return out


# ---- training-mode backward (csrc/bcos_train.hip) --------------------------------------------
def train_scale_bwd(gy2d, y2d, s2d, norm, mode, b, force_pow=False, want_bgrad=False, want_absmax=False, bn=None):
    """(gy * dy/dlin [rows,C], dL/dnorm / norm-denominator [rows], dL/dB_eff [1] or None) of y = s(lin, norm) * lin with s not
    detached (include/bcos_hip.h: bcos_train_scale_bwd).  `bn` = (g, mean or None, coef or None): gy2d is the gradient w.r.t. the output
    of the BatchNormUncentered2d behind the layer; the launch applies the norm's input gradient on the way (bcos_train_scale_bwd_bn)."""
    lib = _l.load()
    rows, Cc = gy2d.shape
    glin = torch.empty_like(gy2d)
    rnorm = torch.empty((rows,), device=gy2d.device, dtype=torch.float32)
    if bn is not None:
        if want_bgrad:
            raise BcosHipError("train_scale_bwd: the fused BatchNorm form has no exponent gradient")
        am = torch.empty((rows,), device=gy2d.device, dtype=torch.int32) if want_absmax else None
        g, mean, coef = bn
        _l.check(lib.bcos_train_scale_bwd_bn(_dev(gy2d, "g_out"), _dev(y2d, "y"), _dev(s2d, "s"), _dev(norm, "norm"), _dev(g, "bn_g"),
                                             _dev(mean, "bn_mean"), _dev(coef, "bn_coef"), _dev(glin, "glin"), _dev(rnorm, "rnorm"),
                                             C.c_void_p(am.data_ptr()) if am is not None else None, rows, Cc, int(mode), float(b),
                                             int(bool(force_pow)), _stream()), "bcos_train_scale_bwd_bn")
        if am is not None:
            _attach_absmax(glin, am)
        return glin, rnorm, None
    bgrad = torch.zeros((1,), device=gy2d.device, dtype=torch.float32) if want_bgrad else None
    am = torch.empty((rows,), device=gy2d.device, dtype=torch.int32) if want_absmax else None      # per-row max |glin| (the reader's operand scale)
    _l.check(lib.bcos_train_scale_bwd_absmax(_dev(gy2d, "gy"), _dev(y2d, "y"), _dev(s2d, "s"), _dev(norm, "norm"), _dev(glin, "glin"),
                                             _dev(rnorm, "rnorm"), _dev(bgrad, "bgrad"), C.c_void_p(am.data_ptr()) if am is not None else None,
                                             rows, Cc, int(mode), float(b), int(bool(force_pow)), _stream()),
             "bcos_train_scale_bwd")
    if am is not None:
        _attach_absmax(glin, am)
    return glin, rnorm, bgrad


def weight_rownorm_bwd(w2d, g2d, gain=None, want_gw=True, want_ggain=False):
    """Backward of weight_rownorm_scale: (gw [rows, cols] or None, ggain [rows] or None), include/bcos_hip.h."""
    lib = _l.load()
    rows, cols = w2d.shape
    gw = torch.empty_like(w2d) if want_gw else None
    gg = torch.empty((rows,), device=w2d.device, dtype=torch.float32) if want_ggain else None
    _l.check(lib.bcos_weight_rownorm_bwd(_dev(w2d, "w"), _dev(g2d, "g_eff"), _dev(gain, "gain"), _dev(gw, "gw"), _dev(gg, "ggain"),
                                         rows, cols, _stream()), "bcos_weight_rownorm_bwd")
    return gw, gg


def maxout_scatter(g2d, argmax2d, max_out):
    """[rows, Cout] gradient w.r.t. the MaxOut units -> [rows, Cout * M]: the value at each unit's winning filter, 0 elsewhere."""
    lib = _l.load()
    rows, Cout = g2d.shape
    full = torch.empty((rows, Cout * max_out), device=g2d.device, dtype=torch.float32)
    if argmax2d.dtype != torch.int32 or not argmax2d.is_cuda or not argmax2d.is_contiguous():
        raise BcosHipError("maxout_scatter: argmax must be a contiguous int32 HIP tensor")
    _l.check(lib.bcos_maxout_scatter(_dev(g2d, "g"), C.c_void_p(argmax2d.data_ptr()), _dev(full, "full"), rows, Cout, int(max_out),
                                     _stream()), "bcos_maxout_scatter")
    return full


def patch_norm_bwd(x, rnorm, C_used, kernel, stride, padding, dilation, addend=None):
    """x [N,H,W,pitch] (first C_used channels), rnorm [N,P,Q] -> x * PatchSum^T(rnorm) (+ addend) [N,H,W,C_used]
    (include/bcos_hip.h: bcos_patch_norm_bwd / bcos_patch_norm_bwd_add)."""
    lib = _l.load()
    N, H, W, pitch = x.shape
    _, P, Q = rnorm.shape
    out = torch.empty((N, H, W, C_used), device=x.device, dtype=torch.float32)
    if addend is not None and tuple(addend.shape) != (N, H, W, C_used):
        raise BcosHipError(f"patch_norm_bwd: addend {tuple(addend.shape)} does not match the output {(N, H, W, C_used)}")
    _l.check(lib.bcos_patch_norm_bwd_add(_dev(x, "x"), _dev(rnorm, "rnorm"), _dev(addend, "addend"), _dev(out, "out"), N, H, W, C_used, pitch,
                                         P, Q, kernel[0], kernel[1], stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1],
                                         _stream()), "bcos_patch_norm_bwd")
    return out


class ZeroArena:
    """The zero-filled accumulators of one backward pass (weight gradients: pixel chunks meet in them through atomics) from ONE fill:
    `begin()` allocates what the previous pass asked for, `take(shape)` hands out the next slice (torch.zeros while the arena is
    missing or exhausted -- the first pass)."""

    def __init__(self):
        self.buf, self.used, self.want = None, 0, 0

    def begin(self, device):
        self.buf = torch.zeros(self.want, device=device, dtype=torch.float32) if self.want else None
        self.used = self.want = 0

    def take(self, shape, device):
        n = 1
        for d in shape:
            n *= int(d)
        n_al = (n + 3) & ~3
        self.want += n_al
        if self.buf is None or self.buf.device != torch.device(device) or self.used + n_al > self.buf.numel():
            return torch.zeros(shape, device=device, dtype=torch.float32)
        out = self.buf[self.used:self.used + n].view(shape)
        self.used += n_al
        return out

    def end(self):
        self.buf = None          # (the slices keep the storage alive for as long as the gradients live)


WGRAD_ORDERED = os.environ.get("BCOS_WGRAD_ORDERED", "1") != "0"      # development A/B: 0 = the round-5 kernel (pixel chunks combined with fp32 atomics)
_WGRAD_WS = {}                     # (device, stream) -> workspace of the ordered weight gradient (launches of one stream run in order: one buffer)


def wgrad_is_ordered() -> bool:
    """Does conv2d_wgrad write its result (fixed-order combine, no zeroed accumulator needed) rather than accumulate into it?"""
    return WGRAD_ORDERED and _l.get_contraction_mode() != "f32"


_WGRAD_NEED = {}                   # geometry -> workspace floats (one library call per distinct layer geometry and option value, not per launch)


def _wgrad_workspace(floats: int, device, stream) -> torch.Tensor:
    key = (device.index, stream.value)
    buf = _WGRAD_WS.get(key)
    if buf is None or buf.numel() < floats:
        buf = torch.empty(max(int(floats), 1 << 22), device=device, dtype=torch.float32)
        _WGRAD_WS[key] = buf
    return buf


def conv2d_wgrad(glin, x, C_used, Cout, kernel, stride, padding, dilation, out=None):
    """glin [N,P,Q,g_pitch] (first Cout channels), x [N,H,W,x_pitch] (first C_used) -> gw [Cout,kh,kw,C_used].
    Default (round 6): include/bcos_hip.h: bcos_conv2d_wgrad_ordered -- both operands split into bf16 planes once at staging, the pixel
    chunks' partial tiles added in a FIXED order through a workspace: bit-identical from call to call; `out` need not be zeroed.
    Contraction mode f32, a gw that is not a multiple of 4 floats, or BCOS_WGRAD_ORDERED=0: bcos_conv2d_wgrad (fp32 atomics; `out`: a
    ZEROED [Cout,kh,kw,C_used])."""
    lib = _l.load()
    N, H, W, x_pitch = x.shape
    _, P, Q, g_pitch = glin.shape
    shape = (Cout, kernel[0], kernel[1], C_used)
    if out is not None and (tuple(out.shape) != shape or not out.is_contiguous()):
        raise BcosHipError(f"conv2d_wgrad: out must be a contiguous {shape} tensor")
    geo = (N, H, W, C_used, x_pitch, P, Q, Cout, g_pitch, kernel[0], kernel[1], stride[0], stride[1], padding[0], padding[1], dilation[0],
           dilation[1], C_used)
    if WGRAD_ORDERED and _l.get_contraction_mode() != "f32" and (Cout * kernel[0] * kernel[1] * C_used) % 4 == 0:
        gw = out if out is not None else torch.empty(shape, device=x.device, dtype=torch.float32)
        nkey = geo + (_l.OPTION_GEN,)           # (BCOS_OPT_WGRAD_WGS changes the chunking: sizes cached per generation of the option table)
        need = _WGRAD_NEED.get(nkey)
        if need is None:
            out_n = C.c_int64(0)
            _l.check(lib.bcos_conv2d_wgrad_ws_floats(*geo, C.byref(out_n)), "bcos_conv2d_wgrad_ws_floats")
            need = _WGRAD_NEED[nkey] = out_n.value
        st = _stream()
        ws = _wgrad_workspace(need, x.device, st) if need else None
        _l.check(lib.bcos_conv2d_wgrad_ordered(_dev(glin, "glin"), _dev(x, "x"), _dev(gw, "gw"), _dev(ws, "ws"), *geo, st),
                 "bcos_conv2d_wgrad_ordered")
        return gw
    gw = out if out is not None else torch.zeros(shape, device=x.device, dtype=torch.float32)
    _l.check(lib.bcos_conv2d_wgrad(_dev(glin, "glin"), _dev(x, "x"), _dev(gw, "gw"), *geo, _stream()), "bcos_conv2d_wgrad")
    return gw


def colsum(a2d, b2d=None, shift_a=None, shift_b=None):
    """out[c] = sum_r (a[r,c] - shift_a[c]) * (b[r,c] - shift_b[c])   (b None: plain column sums); C % 4 == 0.
    Two launches over a torch-owned workspace (include/bcos_hip.h: bcos_colsum_ws): full bandwidth, no atomics, one fixed summation
    order per (rows, C) -- bit-identical from run to run."""
    lib = _l.load()
    rows, Cc = a2d.shape
    n = C.c_int64(0)
    _l.check(lib.bcos_colsum_ws_floats(rows, Cc, C.byref(n)), "bcos_colsum_ws_floats")
    ws = torch.empty((n.value,), device=a2d.device, dtype=torch.float32)
    out = torch.empty((Cc,), device=a2d.device, dtype=torch.float32)
    _l.check(lib.bcos_colsum_ws(_dev(a2d, "a"), _dev(b2d, "b"), _dev(shift_a, "shift_a"), _dev(shift_b, "shift_b"), _dev(out, "out"),
                                _dev(ws, "workspace"), n.value, rows, Cc, _stream()), "bcos_colsum_ws")
    return out


def _bn_ws(rows, Cc, device):
    n = C.c_int64(0)
    _l.check(_l.load().bcos_bn_train_ws_floats(rows, Cc, C.byref(n)), "bcos_bn_train_ws_floats")
    return torch.empty((n.value,), device=device, dtype=torch.float32), n.value


def bn_batch_stats(y2d, weight, eps, running_var=None, momentum=0.0):
    """-> (mean, var, rstd, g = weight * rstd) of a BatchNormUncentered2d in training mode from ONE pass over y [rows, C]
    (include/bcos_hip.h: bcos_bn_batch_stats); `running_var` is updated in place with `momentum`."""
    rows, Cc = y2d.shape
    ws, n = _bn_ws(rows, Cc, y2d.device)
    mean, var, rstd, g = (torch.empty((Cc,), device=y2d.device, dtype=torch.float32) for _ in range(4))
    _l.check(_l.load().bcos_bn_batch_stats(_dev(y2d, "y"), _dev(weight, "weight"), _dev(running_var, "running_var"), _dev(mean, "mean"),
                                           _dev(var, "var"), _dev(rstd, "rstd"), _dev(g, "g"), _dev(ws, "workspace"), n, rows, Cc, float(eps),
                                           float(momentum), _stream()), "bcos_bn_batch_stats")
    if running_var is not None:          # written through its pointer: tell torch (plans keyed on tensor versions re-read it, engine._Conv.fingerprint)
        torch.autograd.graph.increment_version(running_var)
    return mean, var, rstd, g


def relu_bwd_colsums(g2d, act2d, y2d, rstd=None, gvec=None, want_sg=False, want_gw=False, want_coef=False):
    """-> (ga, sgx, sg, gw, coef): the ReLU gate of a unit's gradient and the column sums of its BatchNorm backward from one pass
    (include/bcos_hip.h: bcos_relu_bwd_colsums).  `act2d` None: no gate, ga is g2d itself."""
    rows, Cc = g2d.shape
    ws, n = _bn_ws(rows, Cc, g2d.device)
    new = lambda: torch.empty((Cc,), device=g2d.device, dtype=torch.float32)      # noqa: E731
    ga = torch.empty_like(g2d) if act2d is not None else None
    sgx, sg, gw, coef = new(), (new() if want_sg else None), (new() if want_gw else None), (new() if want_coef else None)
    _l.check(_l.load().bcos_relu_bwd_colsums(_dev(g2d, "g"), _dev(act2d, "act"), _dev(y2d, "y"), _dev(ga, "ga"), _dev(rstd, "rstd"),
                                             _dev(gvec, "gvec"), _dev(sgx, "sgx"), _dev(sg, "sg"), _dev(gw, "gw"), _dev(coef, "coef"),
                                             _dev(ws, "workspace"), n, rows, Cc, _stream()), "bcos_relu_bwd_colsums")
    return (ga if ga is not None else g2d), sgx, sg, gw, coef


def colsum_atomic(a2d, b2d=None, shift_a=None, shift_b=None):
    """The single-launch form (bcos_colsum: partial sums combined with fp32 atomics, order not fixed); kept for A/B timing."""
    lib = _l.load()
    rows, Cc = a2d.shape
    out = torch.zeros((Cc,), device=a2d.device, dtype=torch.float32)
    _l.check(lib.bcos_colsum(_dev(a2d, "a"), _dev(b2d, "b"), _dev(shift_a, "shift_a"), _dev(shift_b, "shift_b"), _dev(out, "out"),
                             rows, Cc, _stream()), "bcos_colsum")
    return out


def colsum_ordered(a2d, b2d=None, shift_a=None, shift_b=None):
    """colsum in a fixed summation order (include/bcos_hip.h: bcos_colsum_ordered): bit-identical across runs and processes."""
    lib = _l.load()
    rows, Cc = a2d.shape
    out = torch.empty((Cc,), device=a2d.device, dtype=torch.float32)
    _l.check(lib.bcos_colsum_ordered(_dev(a2d, "a"), _dev(b2d, "b"), _dev(shift_a, "shift_a"), _dev(shift_b, "shift_b"), _dev(out, "out"),
                                     rows, Cc, _stream()), "bcos_colsum_ordered")
    return out


def channel_moments_ordered(x_nchw: torch.Tensor):
    """(mean [C], biased variance [C], mean of squares over everything) of a HIP tensor [N, C, H, W] (channels-last or not) / [R, C],
    every sum in a fixed order (colsum: bcos_colsum_ws): what torch's x.var((0, 2, 3), unbiased=False) and x.pow(2).mean() give, reproducibly."""
    if x_nchw.dim() == 4:
        x2 = x_nchw.permute(0, 2, 3, 1).contiguous()
        x2 = x2.view(-1, x2.shape[-1])
    else:
        x2 = x_nchw.contiguous().view(-1, x_nchw.shape[-1])
    Cc = x2.shape[1]
    if Cc % 4:
        x2 = torch.nn.functional.pad(x2, (0, 4 - Cc % 4))
    rows = x2.shape[0]
    mean = colsum(x2) / rows
    var = colsum(x2, x2, mean, mean) / rows
    sq = colsum(x2, x2)[:Cc]
    msq = sq.double().cpu().sum() / (rows * Cc)              # (C <= a few thousand values: summed on the host, in order)
    return mean[:Cc], var[:Cc], float(msq)


def channel_axpby(a, sa, b=None, mb=None, sb=None, out=None):
    """out = a * sa[c] + (b - mb[c]) * sb[c] over the last (channel) dimension."""
    lib = _l.load()
    Cc = a.shape[-1]
    if out is None:
        out = torch.empty_like(a)
    _l.check(lib.bcos_channel_axpby(_dev(a, "a"), _dev(sa, "sa"), _dev(b, "b"), _dev(mb, "mb"), _dev(sb, "sb"), _dev(out, "out"),
                                    a.numel() // Cc, Cc, _stream()), "bcos_channel_axpby")
    return out


# ---- transformer pieces (csrc/bcos_vit.hip) ----------------------------------------------------------------
def _am_ptr(am):
    return C.c_void_p(am.data_ptr()) if am is not None else None


def layernorm_fwd(x2d, weight, bias, eps, want_rstd=False, out=None, want_absmax=False):
    """`want_absmax`: the kernel also emits the row maxima of y for the split-f16 contraction that reads it."""
    lib = _l.load()
    rows, D = x2d.shape
    y = out if out is not None else torch.empty_like(x2d)
    rstd = torch.empty((rows,), device=x2d.device, dtype=torch.float32) if want_rstd else None
    am = _fused_absmax(y, want_absmax)
    _l.check(lib.bcos_layernorm_fwd(_dev(x2d, "x"), _dev(weight, "w"), _dev(bias, "b"), _dev(y, "y"), _dev(rstd, "rstd"), _am_ptr(am),
                                    rows, D, float(eps), _stream()), "bcos_layernorm_fwd")
    return y, rstd


def layernorm_stats(x2d, weight, bias, eps, want_zsumsq=False, want_absmax=False):
    """Row statistics of a LayerNorm whose output is never written (include/bcos_hip.h: bcos_layernorm_stats): (rstd, |y|^2 or None);
    `want_absmax`: x2d gets its operand maxima attached (the contraction that follows reads x2d itself)."""
    lib = _l.load()
    rows, D = x2d.shape
    rstd = torch.empty((rows,), device=x2d.device, dtype=torch.float32)
    zss = torch.empty((rows,), device=x2d.device, dtype=torch.float32) if want_zsumsq else None
    am = _fused_absmax(x2d, want_absmax and absmax_of(x2d) is None)
    _l.check(lib.bcos_layernorm_stats(_dev(x2d, "x"), _dev(weight, "w"), _dev(bias, "b"), _dev(rstd, "rstd"), _dev(zss, "zss"), _am_ptr(am),
                                      rows, D, float(eps), _stream()), "bcos_layernorm_stats")
    return rstd, zss


def layernorm_bwd_detached(gy2d, weight, rstd, addend=None, mul2=None, want_out=True, want_out2=False, out=None, want_absmax2=False):
    lib = _l.load()
    rows, D = gy2d.shape
    o = (out if out is not None else torch.empty_like(gy2d)) if want_out else None
    o2 = torch.empty_like(gy2d) if want_out2 else None
    am = _fused_absmax(o2, want_absmax2) if o2 is not None else None
    _l.check(lib.bcos_layernorm_bwd_detached(_dev(gy2d, "gy"), _dev(weight, "w"), _dev(rstd, "rstd"), _dev(addend, "addend"),
                                             _dev(mul2, "mul2"), _dev(o, "out"), _dev(o2, "out2"), _am_ptr(am), rows, D, _stream()),
             "bcos_layernorm_bwd_detached")
    return o, o2


def layernorm_bwd(gy2d, x2d, weight, rstd, want_xhat=False, addend=None):
    """Full LayerNorm input gradient (+ x_hat for the weight gradient; + `addend`, the residual stream's gradient),
    include/bcos_hip.h: bcos_layernorm_bwd_add."""
    lib = _l.load()
    rows, D = gy2d.shape
    gx = torch.empty_like(gy2d)
    xhat = torch.empty_like(gy2d) if want_xhat else None
    if addend is not None and tuple(addend.shape) != (rows, D):
        raise BcosHipError(f"layernorm_bwd: addend {tuple(addend.shape)} does not match {(rows, D)}")
    _l.check(lib.bcos_layernorm_bwd_add(_dev(gy2d, "gy"), _dev(x2d, "x"), _dev(weight, "weight"), _dev(rstd, "rstd"), _dev(addend, "addend"),
                                        _dev(gx, "gx"), _dev(xhat, "xhat"), rows, D, _stream()), "bcos_layernorm_bwd")
    return gx, xhat


def groupnorm_bwd(gy_nhwc, x_nhwc, groups, weight, rstd, want_xhat=False):
    """Full GroupNorm input gradient (+ x_hat for the affine gradients), include/bcos_hip.h: bcos_groupnorm_bwd."""
    lib = _l.load()
    N, H, W, Cc = gy_nhwc.shape
    gx = torch.empty_like(gy_nhwc)
    xhat = torch.empty_like(gy_nhwc) if want_xhat else None
    _l.check(lib.bcos_groupnorm_bwd(_dev(gy_nhwc, "gy"), _dev(x_nhwc, "x"), _dev(weight, "weight"), _dev(rstd, "rstd"), _dev(gx, "gx"),
                                    _dev(xhat, "xhat"), N, H * W, Cc, int(groups), _stream()), "bcos_groupnorm_bwd")
    return gx, xhat


def gelu_bwd(gy, x):
    lib = _l.load()
    gx = torch.empty_like(gy)
    _l.check(lib.bcos_gelu_bwd(_dev(gy, "gy"), _dev(x, "x"), _dev(gx, "gx"), gy.numel(), _stream()), "bcos_gelu_bwd")
    return gx


def attention_bwd(qkv, stats, out, gout, heads, scale):
    """gradient w.r.t. the packed qkv [B,T,3*inner] with q, k and v all differentiated (include/bcos_hip.h: bcos_attention_bwd)"""
    lib = _l.load()
    B, T, three_inner = qkv.shape
    gqkv = torch.empty_like(qkv)
    _l.check(lib.bcos_attention_bwd(_dev(qkv, "qkv"), _dev(stats, "stats"), _dev(out, "out"), _dev(gout, "gout"), _dev(gqkv, "gqkv"),
                                    B, T, heads, three_inner // (3 * heads), float(scale), _stream()), "bcos_attention_bwd")
    return gqkv


def groupnorm_fwd(x_nhwc, groups, weight, bias, eps, want_rstd=False):
    """GroupNorm of an NHWC tensor [N,H,W,C] (include/bcos_hip.h: bcos_groupnorm_fwd) -> (y, rstd [N*G] or None)."""
    lib = _l.load()
    N, H, W, Cc = x_nhwc.shape
    y = torch.empty_like(x_nhwc)
    rstd = torch.empty((N * groups,), device=x_nhwc.device, dtype=torch.float32) if want_rstd else None
    _l.check(lib.bcos_groupnorm_fwd(_dev(x_nhwc, "x"), _dev(weight, "weight"), _dev(bias, "bias"), _dev(y, "y"), _dev(rstd, "rstd"),
                                    N, H * W, Cc, int(groups), float(eps), _stream()), "bcos_groupnorm_fwd")
    return y, rstd


def groupnorm_bwd_detached(gy_nhwc, groups, weight, rstd):
    lib = _l.load()
    N, H, W, Cc = gy_nhwc.shape
    gx = torch.empty_like(gy_nhwc)
    _l.check(lib.bcos_groupnorm_bwd_detached(_dev(gy_nhwc, "gy"), _dev(weight, "weight"), _dev(rstd, "rstd"), _dev(gx, "gx"),
                                             N, H * W, Cc, int(groups), _stream()), "bcos_groupnorm_bwd_detached")
    return gx


def gelu_gate(x, want_gate=False, out=None):
    lib = _l.load()
    y = out if out is not None else torch.empty_like(x)
    gate = torch.empty_like(x) if want_gate else None
    _l.check(lib.bcos_gelu_gate(_dev(x, "x"), _dev(y, "y"), _dev(gate, "gate"), x.numel(), _stream()), "bcos_gelu_gate")
    return y, gate


def add_rows_bcast(x, pe):
    lib = _l.load()
    _l.check(lib.bcos_add_rows_bcast(_dev(x, "x"), _dev(pe, "pe"), x.numel(), pe.numel(), _stream()), "bcos_add_rows_bcast")
    drop_absmax(x)              # modified in place by a kernel: the recorded per-pixel maxima no longer bound it
    return x


def attention_fwd(qkv, heads, scale, want_stats=False, want_absmax=False):
    """qkv [B,T,3*H*64] -> out [B,T,H*64] (+ stats [B,H,T,2]).  `want_absmax`: row maxima of out emitted by the kernel."""
    lib = _l.load()
    B, T, three_inner = qkv.shape
    inner = three_inner // 3
    out = torch.empty((B, T, inner), device=qkv.device, dtype=torch.float32)
    stats = torch.empty((B, heads, T, 2), device=qkv.device, dtype=torch.float32) if want_stats else None
    am = _fused_absmax(out, want_absmax)           # zero-filled: the heads meet in an atomic max
    _l.check(lib.bcos_attention_fwd(_dev(qkv, "qkv"), _dev(out, "out"), _dev(stats, "stats"), _am_ptr(am), B, T, heads, inner // heads,
                                    float(scale), _stream()), "bcos_attention_fwd")
    return out, stats


def attention_bwd_v(qkv, stats, gout, heads, scale, want_absmax=False):
    lib = _l.load()
    B, T, three_inner = qkv.shape
    inner = three_inner // 3
    gv = torch.empty((B, T, inner), device=qkv.device, dtype=torch.float32)
    am = _fused_absmax(gv, want_absmax)
    _l.check(lib.bcos_attention_bwd_v(_dev(qkv, "qkv"), _dev(stats, "stats"), _dev(gout, "gout"), _dev(gv, "gv"), _am_ptr(am), B, T,
                                      heads, inner // heads, float(scale), _stream()), "bcos_attention_bwd_v")
    return gv


def finalize_explanation_patches(gp, x_nchw, std6, patch, add_inverse=False, want_weights=True, want_contrib=True, weights_out=None,
                                 contrib_out=None):
    """gp [N*gh*gw, patch*patch*Cpad] (patch-major input gradient) -> W(x) [N,6,H,W], contribution map [N,H,W].
    `weights_out` / `contrib_out`: write into these tensors (slices of a larger batch) instead of new ones."""
    lib = _l.load()
    N, Cx, H, W = x_nchw.shape
    cpad = gp.shape[-1] // (patch * patch)
    wout = (weights_out if weights_out is not None else torch.empty((N, 6, H, W), device=gp.device, dtype=torch.float32)) if want_weights else None
    cout = (contrib_out if contrib_out is not None else torch.empty((N, H, W), device=gp.device, dtype=torch.float32)) if want_contrib else None
    _l.check(lib.bcos_finalize_explanation_patches(_dev(gp, "gp"), _dev(x_nchw, "x"), _dev(std6, "std"), _dev(wout, "w"),
                                                   _dev(cout, "c"), N, Cx, H, W, patch, cpad, int(add_inverse), _stream()),
             "bcos_finalize_explanation_patches")
    return wout, cout


def render_explanations(x: torch.Tensor, weights: torch.Tensor, smooth: int = 15, alpha_percentile: float = 99.5,
                        want_quantiles: bool = False):
    """Batched gradient_to_image on the device (include/bcos_hip.h: bcos_render_explanations).
    x [N,3|6,H,W], weights [N,6,H,W] -> rgba [N,H,W,4] (and the per-image alpha divisors [N])."""
    lib = _l.load()
    N, Cx, H, W = x.shape
    if tuple(weights.shape) != (N, 6, H, W):
        raise BcosHipError(f"render_explanations: weights {tuple(weights.shape)} do not match input {tuple(x.shape)}")
    rgba = torch.empty((N, H, W, 4), device=x.device, dtype=torch.float32)
    scratch = torch.empty((2, N, H, W), device=x.device, dtype=torch.float32)
    qv = torch.empty((N,), device=x.device, dtype=torch.float32) if want_quantiles else None
    code = lib.bcos_render_explanations(_dev(x, "render.x"), _dev(weights, "render.weights"), _dev(rgba, "render.rgba"),
                                        _dev(scratch, "render.scratch"), _dev(qv, "render.q"), N, Cx, H, W, int(smooth or 0),
                                        float(alpha_percentile) / 100.0, 1 if Cx == 3 else 0, _stream())
    _l.check(code, "bcos_render_explanations")
    return (rgba, qv) if want_quantiles else rgba


def box_filter(maps: torch.Tensor, k: int) -> torch.Tensor:
    """avg_pool2d(maps, k, stride=1, padding=(k-1)//2) of [N,H,W] maps (include/bcos_hip.h: bcos_box_filter)."""
    lib = _l.load()
    N, H, W = maps.shape
    out = torch.empty_like(maps)
    _l.check(lib.bcos_box_filter(_dev(maps, "box_filter.in"), _dev(out, "box_filter.out"), N, H, W, int(k), _stream()),
             "bcos_box_filter")
    return out


def localisation_fractions(attr: torch.Tensor, cell_h: int, cell_w: int, neg: bool = False) -> torch.Tensor:
    """attr [T,H,W] -> [T, cells] share of positive attribution per grid cell (bcos_localisation_fractions)."""
    lib = _l.load()
    T, H, W = attr.shape
    cells = (H // cell_h) * (W // cell_w)
    out = torch.empty((T, cells), device=attr.device, dtype=torch.float32)
    _l.check(lib.bcos_localisation_fractions(_dev(attr, "localisation.attr"), _dev(out, "localisation.frac"), T, H, W,
                                             int(cell_h), int(cell_w), 1 if neg else 0, _stream()),
             "bcos_localisation_fractions")
    return out
