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


class _HipMemcpy:
    """device-to-device hipMemcpyAsync by raw pointer (engine-owned memory: BatchNorm sums, momentum buffers)."""
    _hip = None

    @classmethod
    def copy(cls, dst: int, src: int, nbytes: int, stream: int):
        if cls._hip is None:
            cls._hip = C.CDLL("libamdhip64.so")
            cls._hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
            cls._hip.hipMemcpyAsync.restype = C.c_int
        rc = cls._hip.hipMemcpyAsync(C.c_void_p(dst), C.c_void_p(src), nbytes, 3, C.c_void_p(stream))
        if rc != 0:
            raise RuntimeError(f"hipMemcpyAsync failed with {rc}")


def to_c_config(cfg: LSegConfig, img_h: int, img_w: int, max_batch: int, max_labels: int,
                image_dtype: str = "bf16", full_text_context: bool = False, exact_head_grad: bool = False,
                batch_invariant: bool = False, deterministic: bool = False) -> _lib.LSegConfigC:
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
    # "strict": split-precision validation mode ((hi, lo) fp16 operand pairs, ~21 mantissa bits; include/lseg_hip.h)
    c.image_dtype = {"bf16": _lib.LSEG_BF16, "fp16": _lib.LSEG_F16, "strict": _lib.LSEG_F16_SPLIT}[image_dtype]
    c.flags = ((1 if full_text_context else 0) | (2 if exact_head_grad else 0) | (4 if batch_invariant else 0)
               | (8 if deterministic else 0))
    return c


class HipEngine:
    """One engine = one (backbone, H, W, max_batch, max_labels) plan on one GPU."""

    def __init__(self, cfg: LSegConfig, img_h: int, img_w: int, max_batch: int, max_labels: int,
                 device: Optional[torch.device] = None, image_dtype: str = "bf16",
                 full_text_context: bool = False, exact_head_grad: bool = False, batch_invariant: bool = False,
                 deterministic: Optional[bool] = None):
        """deterministic: the training step's column sums (bias gradients, BatchNorm batch statistics) in a fixed order instead of
        fp32 atomics -- the same step twice gives bit-identical gradients (include/lseg_hip.h, flags bit 3).  None = the environment's
        LSEG_DETERMINISTIC, default ON: it costs nothing measurable (tools/train_bench.py, B = 8, two interleaved rounds: 41.0 / 41.2 ms
        with it, 41.2 / 41.2 ms on the atomics -- profiles/r05_train_bench.txt); LSEG_DETERMINISTIC=0 / deterministic=False = atomics."""
        import os
        if deterministic is None:
            deterministic = os.environ.get("LSEG_DETERMINISTIC", "1") not in ("", "0")
        self.deterministic = bool(deterministic)
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.LSegError(-2, "no GPU visible: the LSeg HIP engine has no CPU fallback")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.cfg = cfg
        self.img_h, self.img_w = img_h, img_w
        self.max_batch, self.max_labels = max_batch, max_labels
        self._c = to_c_config(cfg, img_h, img_w, max_batch, max_labels, image_dtype, full_text_context, exact_head_grad, batch_invariant,
                               self.deterministic)
        h = C.c_void_p()
        _lib.check(self.lib.lseg_create(C.byref(self._c), self.device.index or 0, C.byref(h)))
        self._h = h
        self._K = 0
        self._group = 0
        self._keep: List[torch.Tensor] = []
        self.training = False
        self.image_dtype = image_dtype
        self.grads: Dict[str, torch.Tensor] = {}
        self._cb_error: Optional[BaseException] = None     # first exception raised inside a ctypes callback (ctypes would swallow it)

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
        """Bind every tensor of a (reference-layout) state dict and repack on the device.  Tensors already on the engine's device
        are bound IN PLACE (no copy): they are the fp32 masters the fused SGD step updates and the BatchNorm running statistics
        train mode writes (include/lseg_hip.h); `self.bound[key]` is the device tensor behind each key."""
        st = _stream_ptr(self.device)
        self.bound: Dict[str, torch.Tensor] = {}
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
            self.bound[k] = t
            shape = (C.c_int64 * max(1, t.dim()))(*t.shape)
            _lib.check(self.lib.lseg_bind_param(self._h, k.encode(), C.c_void_p(t.data_ptr()), _DT[t.dtype],
                                                shape, t.dim()))
        _lib.check(self.lib.lseg_finalize_params(self._h, C.c_void_p(st)))

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

    def set_text_features(self, feat: torch.Tensor):
        """fp16 [K, out_c] text features computed elsewhere (un-normalised `encode_text` output or already normalised): replaces the
        tokens; the engine's text tower is not run until set_tokens is called again (lseg_set_text_features)."""
        f = feat.detach().to(self.device, torch.float16).contiguous()
        if f.dim() != 2 or f.shape[1] != self.cfg.out_c:
            raise ValueError(f"text features must be [K, {self.cfg.out_c}], got {tuple(f.shape)}")
        _lib.check(self.lib.lseg_set_text_features(self._h, C.c_void_p(f.data_ptr()), f.shape[0], C.c_void_p(_stream_ptr(self.device))))
        self._keep = [f]
        self._K = f.shape[0]
        self._group = 0

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
        amax = torch.empty((B, H, W), dtype=torch.uint8, device=self.device) if want_argmax else None      # the masks
        _lib.check(self.lib.lseg_forward(
            self._h, C.c_void_p(x.data_ptr()), B,
            C.c_void_p(logits.data_ptr()) if logits is not None else None,
            C.c_void_p(amax.data_ptr()) if amax is not None else None,
            C.c_void_p(_stream_ptr(self.device))))
        self._raise_callback_error()
        if want_logits and want_argmax:
            return logits, amax
        return logits if want_logits else amax

    # ---- training step (include/lseg_hip.h "training step"; modules/lsegmentation_module.py:66-81) ---------------------
    def trainable_keys(self, sd: Dict[str, torch.Tensor]) -> List[str]:
        """State-dict keys (relative to `net.`) the backward produces gradients for, in bucket order."""
        keys = []
        for k, v in sd.items():
            if not v.is_floating_point() or v.dtype != torch.float32 or k.endswith(("running_mean", "running_var")):
                continue
            if k.startswith(_SKIP_PREFIX) or ".refinenet4.resConfUnit1." in k or k.startswith("clip_pretrained."):
                continue
            b = self.lib.lseg_grad_bucket(self._h, k.encode())
            if b >= 0:
                keys.append((b, k))
        return [k for _, k in sorted(keys, key=lambda t: t[0])]

    def enable_training(self, sd: Dict[str, torch.Tensor]):
        """net.train(): allocate the saved-activation workspace, and one FLAT fp32 gradient buffer per bucket (what the RCCL
        all-reduce runs on, in place); every parameter's gradient is a view into its bucket, bound to the engine."""
        _lib.check(self.lib.lseg_set_train(self._h, 1))
        nb = self.lib.lseg_num_grad_buckets(self._h)
        groups: List[List[str]] = [[] for _ in range(nb)]
        for k in self.trainable_keys(sd):
            groups[self.lib.lseg_grad_bucket(self._h, k.encode())].append(k)
        self.grad_buckets: List[torch.Tensor] = []
        self.grads = {}
        for ks in groups:
            n = sum(sd[k].numel() for k in ks)
            flat = torch.zeros(max(n, 1), dtype=torch.float32, device=self.device)
            off = 0
            for k in ks:
                m = sd[k].numel()
                view = flat[off:off + m].view(sd[k].shape)
                _lib.check(self.lib.lseg_bind_grad(self._h, k.encode(), C.c_void_p(view.data_ptr())))
                self.grads[k] = view
                off += m
            self.grad_buckets.append(flat)
        self._loss = torch.zeros(2, dtype=torch.float64, device=self.device)
        self.training = True

    def set_train(self, enabled: bool):
        _lib.check(self.lib.lseg_set_train(self._h, int(enabled)))
        self.training = bool(enabled)

    def _raise_callback_error(self):
        """Exceptions raised inside the bucket / SyncBatchNorm callbacks cannot cross the C frames (ctypes prints and drops them):
        the trampolines park the first one here and it is re-raised as soon as the C call that triggered it has returned."""
        if self._cb_error is not None:
            e, self._cb_error = self._cb_error, None
            raise RuntimeError("a gradient-bucket / SyncBatchNorm callback failed during the engine call; gradients or batch "
                               "statistics of this step are NOT exchanged") from e

    def backward(self, target: Optional[torch.Tensor] = None, dlogits: Optional[torch.Tensor] = None,
                 ignore_index: int = -1, accumulate: bool = False, grad_scale: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """After a train-mode forward.  target int64 [B,H,W] -> returns the mean cross-entropy (0-dim tensor on the device,
        no host sync); or dlogits fp32 [B,K,H,W] (autograd hand-over) -> None.  Gradients land in self.grads / grad_buckets.
        grad_scale (fp32 0-dim device tensor, target form only): d(loss) of an autograd caller, read by the kernel (lseg_backward_scaled)."""
        st = C.c_void_p(_stream_ptr(self.device))
        if dlogits is not None:
            dl = dlogits.contiguous()
            assert dl.dtype == torch.float32 and dl.device == self.device
            _lib.check(self.lib.lseg_backward(self._h, C.c_void_p(dl.data_ptr()), None, ignore_index, int(accumulate), None, st))
            self._raise_callback_error()
            return None
        t = target.contiguous()
        assert t.dtype == torch.int64 and t.device == self.device
        if grad_scale is not None:
            gs = grad_scale.detach().to(self.device, torch.float32).reshape(1).contiguous()
            self._keep = [t, gs]                                  # alive until the kernels that read them have been enqueued AND run
            _lib.check(self.lib.lseg_backward_scaled(self._h, C.c_void_p(t.data_ptr()), ignore_index, int(accumulate),
                                                     C.c_void_p(gs.data_ptr()), st))
            self._raise_callback_error()
            return None
        _lib.check(self.lib.lseg_backward(self._h, None, C.c_void_p(t.data_ptr()), ignore_index, int(accumulate),
                                          C.c_void_p(self._loss.data_ptr()), st))
        self._raise_callback_error()
        return (self._loss[0] / self._loss[1]).float()

    def train_loss(self, target: torch.Tensor, ignore_index: int = -1, want_counts: bool = False):
        """Value of the criterion on the last train-mode forward without the backward (lseg_train_loss): the mean cross-entropy as a
        0-dim fp32 device tensor (no host sync) and, optionally, int64[2] = {correct, labeled} of the arg-max mask."""
        t = target.contiguous()
        assert t.dtype == torch.int64 and t.device == self.device
        self._loss_target = t                                     # the engine remembers the pointer for the following backward
        nll = torch.empty(2, dtype=torch.float64, device=self.device)
        cnt = torch.empty(2, dtype=torch.int64, device=self.device) if want_counts else None
        _lib.check(self.lib.lseg_train_loss(self._h, C.c_void_p(t.data_ptr()), ignore_index, C.c_void_p(nll.data_ptr()),
                                            C.c_void_p(cnt.data_ptr()) if cnt is not None else None,
                                            C.c_void_p(_stream_ptr(self.device))))
        loss = (nll[0] / nll[1]).float()
        return (loss, cnt) if want_counts else loss

    # ---- momentum buffers of the fused SGD (torch.optim.SGD state['momentum_buffer']; checkpoints, engine rebuilds) ----
    def _momentum_ptr(self, key: str):
        p, n = C.c_void_p(), C.c_size_t(0)
        _lib.check(self.lib.lseg_sgd_momentum(self._h, key.encode(), C.byref(p), C.byref(n)))
        return p.value, n.value

    def get_momentum(self, key: str) -> torch.Tensor:
        ptr, n = self._momentum_ptr(key)
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        _HipMemcpy.copy(out.data_ptr(), ptr, 4 * n, _stream_ptr(self.device))
        return out.view(self.bound[key].shape)

    def set_momentum(self, key: str, value: torch.Tensor):
        ptr, n = self._momentum_ptr(key)
        v = value.detach().to(self.device, torch.float32).contiguous()
        if v.numel() != n:
            raise ValueError(f"momentum of '{key}' has {v.numel()} elements, expected {n}")
        _HipMemcpy.copy(ptr, v.data_ptr(), 4 * n, _stream_ptr(self.device))
        torch.cuda.current_stream(self.device).synchronize()       # v may be a temporary

    def mark_sgd_initialized(self, initialized: bool = True):
        _lib.check(self.lib.lseg_sgd_mark_initialized(self._h, int(initialized)))

    def sgd_step(self, lr_pretrained: float, lr_scratch: float, momentum: float = 0.9, weight_decay: float = 1e-4):
        """Fused SGD on the bound fp32 masters + re-pack of the engine's operand copies.  The masters (the caller's tensors) and the
        BatchNorm running statistics are written through raw pointers: tensor._version does NOT move -- other consumers of the
        same tensors (other engines of an LSegNet) must be told (LSeg.invalidate_engines)."""
        _lib.check(self.lib.lseg_sgd_step(self._h, lr_pretrained, lr_scratch, momentum, weight_decay,
                                          C.c_void_p(_stream_ptr(self.device))))

    def _guarded(self, fn):
        def call(*a):
            try:
                fn(*a)
            except BaseException as e:          # noqa: BLE001  (must not unwind through the C frames; re-raised by _raise_callback_error)
                if self._cb_error is None:
                    self._cb_error = e
        return call

    def set_bucket_callback(self, fn):
        """fn(bucket_index) is called on the host as soon as the bucket's last gradient kernel is enqueued."""
        g = self._guarded(fn) if fn is not None else None
        self._bucket_cb = _lib.BUCKET_CB(lambda user, b, stream: g(int(b))) if fn is not None else None
        _lib.check(self.lib.lseg_set_bucket_callback(self._h, C.cast(self._bucket_cb, C.c_void_p) if fn else None, None))

    def set_bn_sync(self, fn, world_size: int):
        """fn(dev_ptr, n_floats) must sum the n floats at dev_ptr over the ranks, ordered on the current stream."""
        g = self._guarded(fn) if fn is not None else None
        self._bn_cb = _lib.REDUCE_CB(lambda user, p, n, stream: g(int(p), int(n))) if fn is not None else None
        _lib.check(self.lib.lseg_set_bn_sync(self._h, C.cast(self._bn_cb, C.c_void_p) if fn else None, None, int(world_size)))

    def forward_stats(self, target: torch.Tensor, ignore_index: int = -1) -> dict:
        """pixAcc / IoU counts + CE sum of the last forward's output vs `target` int64 [B,H,W] (lseg_forward_stats): computed from the
        engine's low-resolution logits through the x2 bilinear on the fly -- pair it with forward(x, want_logits=False)."""
        t = target.detach().to(self.device, torch.int64).contiguous()
        K = self._group if self._group > 0 else self._K
        counts = torch.empty(2 + 3 * K, dtype=torch.int64, device=self.device)
        nll = torch.empty(2, dtype=torch.float64, device=self.device)
        _lib.check(self.lib.lseg_forward_stats(self._h, C.c_void_p(t.data_ptr()), int(ignore_index), C.c_void_p(counts.data_ptr()),
                                               C.c_void_p(nll.data_ptr()), C.c_void_p(_stream_ptr(self.device))))
        c, n = counts.cpu(), nll.cpu()
        inter, pred, lab = c[2:2 + K], c[2 + K:2 + 2 * K], c[2 + 2 * K:2 + 3 * K]
        return {"correct": int(c[0]), "labeled": int(c[1]), "area_inter": inter, "area_pred": pred, "area_lab": lab,
                "area_union": pred + lab - inter, "nll_sum": float(n[0]), "nll_count": int(n[1])}

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

    def check_range(self) -> dict:
        """16-bit range check of the image tower (lseg_check_range): non-finite values, values within a factor 2 of the fp16 limit and
        the largest finite magnitude over every 16-bit activation buffer, as the forwards so far left them.  Synchronises."""
        import struct
        out = (C.c_uint64 * 4)()
        _lib.check(self.lib.lseg_check_range(self._h, out, C.c_void_p(_stream_ptr(self.device))))
        return {"nonfinite": int(out[0]), "near_fp16_limit": int(out[1]),
                "max_abs": struct.unpack("f", struct.pack("I", int(out[2]) & 0xffffffff))[0], "scanned": int(out[3])}

    def overflow_seen(self, reset: bool = False) -> bool:
        """True once a COMPLETED inference forward left inf / NaN in the head feature map (lseg_overflow_seen: no synchronisation; an overflow on
        one call is reported at the latest by the next one).  Sticky until reset."""
        r = self.lib.lseg_overflow_seen(self._h, int(reset))
        if r < 0:
            _lib.check(r)
        return bool(r)

    PROFILE_FAMILIES = ("forward", "mlp_fc1", "mlp_fc2", "attn_proj", "attn_qkv", "attention", "layernorm", "correlation")

    def set_profiling(self, enabled):
        """True = HIP-event timing of the whole forward + the MLP fc1 GEMM; a list of family names (PROFILE_FAMILIES) = exactly those;
        False = off.  The events are created here, outside any timed region (lseg_set_profiling)."""
        if isinstance(enabled, (list, tuple, set)):
            mask = 1 << 30               # an unused family bit: keeps a one-family mask apart from the ABI's "1 = forward + mlp_fc1"
            for f in enabled:
                mask |= 1 << self.PROFILE_FAMILIES.index(f)
        else:
            mask = int(bool(enabled))
        _lib.check(self.lib.lseg_set_profiling(self._h, mask))

    def profile(self, family: str):
        ms, n, fl = C.c_double(0), C.c_int64(0), C.c_double(0)
        _lib.check(self.lib.lseg_get_profile(self._h, family.encode(), C.byref(ms), C.byref(n), C.byref(fl)))
        return {"total_ms": ms.value, "launches": n.value, "flops_per_launch": fl.value}
