"""Host-side driver of the HIP engine: owns one C handle per device and feeds it torch
tensors (PyTorch is only the container for HBM allocations and the stream).

Mirrors what `LSeg.forward` (modules/models/lseg_net.py:160-205) needs from the
device side: parameters under the reference's state-dict keys, int64 CLIP tokens,
an fp32 NCHW image batch in, fp32 [B,K,H,W] logits out.
"""
import ctypes as C
from typing import Dict, Iterable, List, Optional, Sequence

import torch

from . import _lib
from .config import LSegConfig

_DT = {torch.float32: _lib.LSEG_F32, torch.float16: _lib.LSEG_F16, torch.bfloat16: _lib.LSEG_BF16,
       torch.int64: _lib.LSEG_I64}
_ACT = {"relu": 0, "lrelu": 1, "tanh": 2}
_RS = {"id": _lib.RS_IDENTITY, "convT": _lib.RS_CONVT, "conv_s2": _lib.RS_CONV_S2}

# state-dict prefixes the engine consumes (everything else, e.g. clip_pretrained.visual.*,
# pretrained.model.norm/head, num_batches_tracked, is host-only baggage)
_SKIP_SUFFIX = ("num_batches_tracked",)
_SKIP_PREFIX = ("clip_pretrained.visual.", "clip_pretrained.logit_scale", "pretrained.model.norm.",
                "pretrained.model.head.")


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def to_c_config(cfg: LSegConfig, img_h: int, img_w: int, max_batch: int, max_labels: int,
                image_dtype: str = "bf16", full_text_context: bool = False) -> _lib.LSegConfigC:
    c = _lib.LSegConfigC()
    c.abi_version = _lib.ABI_VERSION
    c.patch, c.dim, c.depth, c.heads = cfg.patch, cfg.dim, cfg.depth, cfg.heads
    for i in range(4):
        c.hooks[i] = cfg.hooks[i]
        c.reassemble_ch[i] = cfg.reassemble[i]
        c.resample_kind[i] = _RS[cfg.resample[i][0]]
        c.resample_k[i] = cfg.resample[i][1]
    c.pos_grid = cfg.pos_grid
    c.features, c.out_c = cfg.features, cfg.out_c
    c.arch_option, c.block_depth = cfg.arch_option, cfg.block_depth
    c.activation = _ACT[cfg.activation]
    t = cfg.text
    c.text_vocab, c.text_ctx, c.text_width, c.text_heads, c.text_layers = t.vocab, t.ctx, t.width, t.heads, t.layers
    c.img_h, c.img_w = img_h, img_w
    c.max_batch, c.max_labels = max_batch, max_labels
    c.image_dtype = {"bf16": _lib.LSEG_BF16, "fp16": _lib.LSEG_F16}[image_dtype]
    c.flags = 1 if full_text_context else 0
    return c


class HipEngine:
    """One engine = one (backbone, H, W, max_batch, max_labels) plan on one GPU."""

    def __init__(self, cfg: LSegConfig, img_h: int, img_w: int, max_batch: int, max_labels: int,
                 device: Optional[torch.device] = None, image_dtype: str = "bf16",
                 full_text_context: bool = False):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.LSegError(-2, "no GPU visible: the LSeg HIP engine has no CPU fallback")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.cfg = cfg
        self.img_h, self.img_w = img_h, img_w
        self.max_batch, self.max_labels = max_batch, max_labels
        self._c = to_c_config(cfg, img_h, img_w, max_batch, max_labels, image_dtype, full_text_context)
        h = C.c_void_p()
        _lib.check(self.lib.lseg_create(C.byref(self._c), self.device.index or 0, C.byref(h)))
        self._h = h
        self._K = 0
        self._group = 0
        self._keep: List[torch.Tensor] = []

    def close(self):
        if getattr(self, "_h", None):
            self.lib.lseg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters -------------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor], prefix: str = ""):
        """Bind every tensor of a (reference-layout) state dict and repack on the device."""
        st = _stream_ptr(self.device)
        keep = []
        for k, v in sd.items():
            if prefix:
                if not k.startswith(prefix):
                    continue
                k = k[len(prefix):]
            if k.endswith(_SKIP_SUFFIX) or k.startswith(_SKIP_PREFIX):
                continue
            if v.dtype not in _DT:
                continue
            t = v.detach().to(self.device).contiguous()
            keep.append(t)
            shape = (C.c_int64 * max(1, t.dim()))(*t.shape)
            _lib.check(self.lib.lseg_bind_param(self._h, k.encode(), C.c_void_p(t.data_ptr()), _DT[t.dtype],
                                                shape, t.dim()))
        _lib.check(self.lib.lseg_finalize_params(self._h, C.c_void_p(st)))
        del keep      # finalize synchronises: the engine now owns packed copies

    # ---- text -------------------------------------------------------------------------------------------
    def set_tokens(self, tokens: torch.Tensor, labels_per_image: int = 0):
        """tokens int64 [K, ctx].  labels_per_image = k > 0 (LSegNetZS, lseg_net_zs.py:177-214): the K = B*k rows are
        per-image label sets, image b is correlated with rows [b*k, (b+1)*k) and forward returns [B, k, H, W]."""
        tok = tokens.detach().to("cpu", torch.int64).contiguous()
        K, ctx = tok.shape
        arr = (C.c_int64 * (K * ctx)).from_buffer_copy(tok.numpy().tobytes())
        _lib.check(self.lib.lseg_set_text_tokens(self._h, arr, K, ctx))
        _lib.check(self.lib.lseg_set_text_grouping(self._h, int(labels_per_image)))
        self._K = K
        self._group = int(labels_per_image)

    def encode_text(self) -> torch.Tensor:
        _lib.check(self.lib.lseg_encode_text(self._h, C.c_void_p(_stream_ptr(self.device))))
        out = torch.empty((self._K, self.cfg.out_c), dtype=torch.float16, device=self.device)
        _lib.check(self.lib.lseg_get_text_features(self._h, C.c_void_p(out.data_ptr()),
                                                   C.c_void_p(_stream_ptr(self.device))))
        return out

    def set_text_cache(self, enabled: bool):
        _lib.check(self.lib.lseg_set_text_cache(self._h, int(enabled)))

    # ---- forward -----------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, want_logits: bool = True, want_argmax: bool = False):
        if x.device != self.device or x.dtype != torch.float32:
            raise ValueError(f"x must be float32 on {self.device}, got {x.dtype} on {x.device}")
        x = x.contiguous()
        B, Cc, H, W = x.shape
        if (Cc, H, W) != (3, self.img_h, self.img_w):
            raise ValueError(f"engine was planned for 3x{self.img_h}x{self.img_w}, got {Cc}x{H}x{W}")
        Kout = self._group if self._group > 0 else self._K
        if self._group > 0 and self._K != B * self._group:
            raise ValueError(f"per-image label sets: {self._K} token rows != batch {B} x {self._group}")
        logits = torch.empty((B, Kout, H, W), dtype=torch.float32, device=self.device) if want_logits else None
        amax = torch.empty((B, H // 2, W // 2), dtype=torch.uint8, device=self.device) if want_argmax else None
        _lib.check(self.lib.lseg_forward(
            self._h, C.c_void_p(x.data_ptr()), B,
            C.c_void_p(logits.data_ptr()) if logits is not None else None,
            C.c_void_p(amax.data_ptr()) if amax is not None else None,
            C.c_void_p(_stream_ptr(self.device))))
        if want_logits and want_argmax:
            return logits, amax
        return logits if want_logits else amax

    # ---- taps / measurement ----------------------------------------------------------------------------
    def set_debug(self, enabled: bool):
        _lib.check(self.lib.lseg_set_debug(self._h, int(enabled)))

    def intermediate(self, name: str, shape: Sequence[int]) -> torch.Tensor:
        out = torch.empty(tuple(shape), dtype=torch.float32, device=self.device)
        n = C.c_size_t(0)
        _lib.check(self.lib.lseg_get_intermediate(self._h, name.encode(), C.c_void_p(out.data_ptr()),
                                                  out.numel(), C.byref(n), C.c_void_p(_stream_ptr(self.device))))
        if n.value != out.numel():
            raise ValueError(f"intermediate '{name}' has {n.value} elements, expected {out.numel()} {tuple(shape)}")
        return out

    def set_profiling(self, enabled: bool):
        _lib.check(self.lib.lseg_set_profiling(self._h, int(enabled)))

    def profile(self, family: str):
        ms, n, fl = C.c_double(0), C.c_int64(0), C.c_double(0)
        _lib.check(self.lib.lseg_get_profile(self._h, family.encode(), C.byref(ms), C.byref(n), C.byref(fl)))
        return {"total_ms": ms.value, "launches": n.value, "flops_per_launch": fl.value}
