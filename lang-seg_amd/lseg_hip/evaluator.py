"""Multi-scale + flip sliding-window evaluator -- the direct caller of the hot path
(reference: additional_utils/encoding_models.py:54-139 `MultiEvalModule.forward` and
additional_utils/models.py:55-140 `LSeg_MultiEvalModule.forward`).

Same algorithm, same arithmetic order per pixel, but every crop of a scale and its mirrored twin
go through the engine as ONE batch instead of 2 x n_crops separate B=1 forwards (a 4:3 ADE image
costs the reference 36 B=1 forwards, each re-encoding the 150 labels; here 6 batched forwards).
Images of a batch are independent in the engine, so the result is identical to the sequential
schedule.  The data movement around the forwards -- resize / pad / crop / flip / overlap-add / count-normalise -- runs on
the device in three kernels per scale (csrc/evaluator.hip through the C ABI: lseg_op_eval_make_crops / _accumulate /
_resize) when the image lives on the GPU; the torch statement of the same steps below is the definition they are tested
against and what runs for host tensors (tests/test_evaluator_ref_golden.py drives it with a toy module on the CPU).
"""
import ctypes as C
import math
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F


def scale_geometry(h: int, w: int, scale: float, base_size: int, crop_size: int):
    """Sizes of one scale of MultiEvalModule.forward (encoding_models.py:66-99): the resized image (height, width), the padded image
    the sliding window runs over (ph, pw), the crop grid (h_grids, w_grids) and the stride.  One crop when the long side fits."""
    stride = int(crop_size * 2.0 / 3.0)
    long_size = int(math.ceil(base_size * scale))
    if h > w:
        height, width = long_size, int(1.0 * w * long_size / h + 0.5)
    else:
        width, height = long_size, int(1.0 * h * long_size / w + 0.5)
    ph, pw = max(height, crop_size), max(width, crop_size)            # pad_image pads up to the crop size, never beyond
    if long_size <= crop_size:
        h_grids = w_grids = 1
    else:
        h_grids = int(math.ceil(1.0 * (ph - crop_size) / stride)) + 1
        w_grids = int(math.ceil(1.0 * (pw - crop_size) / stride)) + 1
    return height, width, ph, pw, h_grids, w_grids, stride


def _pad_image(img, mean, std, crop_size):          # encoding_models.py:144-155
    b, c, h, w = img.shape
    padh = crop_size - h if h < crop_size else 0
    padw = crop_size - w if w < crop_size else 0
    if padh == 0 and padw == 0:
        return img
    pad_values = -np.array(mean) / np.array(std)
    out = img.new_empty((b, c, h + padh, w + padw))
    for i in range(c):
        out[:, i] = F.pad(img[:, i], (0, padw, 0, padh), value=float(pad_values[i]))
    return out


class ModuleSurface(torch.nn.Module):
    """The part of LSegModule's surface the evaluators use (modules/lseg_module.py:29-93: evaluate / evaluate_random, base_size, crop_size,
    mean, std, _up_kwargs) around a bare LSegNet -- for callers that hold a network and no Lightning module (bench.py, tools)."""

    def __init__(self, net, crop_size=480, base_size=520, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
        super().__init__()
        self.net, self.crop_size, self.base_size = net, crop_size, base_size
        self.mean, self.std = list(mean), list(std)
        self._up_kwargs = {"mode": "bilinear", "align_corners": True}

    def evaluate(self, x, target=None):
        return self.net(x)

    def evaluate_random(self, x, labelset, target=None):
        return self.net(x, labelset)


class BatchedMultiEval(torch.nn.Module):
    """Drop-in for `MultiEvalModule(module, nclass, scales=..., flip=...)` on one GPU.

    `module` needs .evaluate(x) / .evaluate_random(x, labels), .base_size, .crop_size, .mean, .std,
    ._up_kwargs (the LSegModule surface, modules/lseg_module.py:29-93)."""

    def __init__(self, module, nclass, flip=True, scales=(0.5, 0.75, 1.0, 1.25, 1.5, 1.75), max_batch=36, cache_text=True):
        """cache_text: encode a label set ONCE per evaluator call chain instead of once per crop batch (the reference re-runs the CLIP text
        tower inside every one of its ~36 forwards per image, lseg_net.py:183 -- the same tokens through the same frozen weights: exact).
        Applied to the wrapped network (`module.net`, an LSegNet) for the duration of a forward and restored afterwards; the engine
        re-encodes whenever the tokens or the weights change (LSegNet._set_tokens / _stamp)."""
        super().__init__()
        self.cache_text = cache_text
        self.module = module
        self.nclass = nclass
        self.base_size = module.base_size
        self.crop_size = module.crop_size
        self.scales = list(scales)
        self.flip = flip
        self.max_batch = max_batch

    def _infer(self, crops: torch.Tensor, label_set) -> torch.Tensor:
        """module_inference (encoding_models.py:133-139) for a stack of crops: out = f(x) + flip(f(flip(x)))."""
        xs = torch.cat([crops, torch.flip(crops, dims=[3])], dim=0) if self.flip else crops
        outs = []
        for i in range(0, xs.shape[0], self.max_batch):
            chunk = xs[i:i + self.max_batch].contiguous()
            outs.append(self.module.evaluate(chunk) if label_set is None
                        else self.module.evaluate_random(chunk, label_set))
        out = torch.cat(outs, dim=0)
        if self.flip:
            n = crops.shape[0]
            out = out[:n] + torch.flip(out[n:], dims=[3])
        return out

    def parallel_forward(self, inputs: Sequence[torch.Tensor], label_set=None) -> List[torch.Tensor]:
        """list of [3,h,w] images in, list of [1,nclass,h,w] score maps out (encoding_models.py:35-52)."""
        return [self.forward(img.unsqueeze(0).cuda(), label_set) for img in inputs]

    def _module_eval(self, xs: torch.Tensor, label_set) -> torch.Tensor:
        outs = []
        for i in range(0, xs.shape[0], self.max_batch):
            chunk = xs[i:i + self.max_batch]
            outs.append(self.module.evaluate(chunk) if label_set is None else self.module.evaluate_random(chunk, label_set))
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)

    @torch.no_grad()
    def _forward_device(self, image: torch.Tensor, label_set=None) -> torch.Tensor:
        """The same schedule with the data movement in csrc/evaluator.hip (image on the GPU)."""
        from . import _lib
        lib = _lib.load()
        P = lambda t: C.c_void_p(t.data_ptr())
        st = C.c_void_p(torch.cuda.current_stream(image.device).cuda_stream)
        batch, ch, h, w = image.shape
        assert batch == 1 and image.dtype == torch.float32
        nclass = self.nclass if label_set is None else len(label_set)
        crop = self.crop_size
        if self.module._up_kwargs != {"mode": "bilinear", "align_corners": True}:
            raise ValueError("the device evaluator implements the reference's bilinear / align_corners=True resize")
        pad = (C.c_float * 3)(*[float(v) for v in (-np.array(self.module.mean) / np.array(self.module.std))])
        image = image.contiguous()
        scores = image.new_zeros((1, nclass, h, w))
        # EVERY crop of EVERY scale is crop x crop: all of them (and their mirrored twins) go through the network as ONE stack, in chunks of
        # max_batch -- a 4:3 ADE image is 36 crops = one batch-36 forward where the per-scale form ran six forwards of 2 .. 12 crops, each at
        # the low-occupancy end of the engine's batch sweep (DESIGN par. 3.8).  Crops are independent in the engine, so the per-crop logits
        # are those of the per-scale form (bit for bit under the batch-invariant schedule).
        geo, stacks = [], []
        for scale in self.scales:
            height, width, ph, pw, h_grids, w_grids, stride = scale_geometry(h, w, scale, self.base_size, crop)
            cur = image.new_empty((ch, height, width))
            _lib.check(lib.lseg_op_eval_resize(P(image), P(cur), ch, h, w, height, width, 0, st))
            n = h_grids * w_grids
            crops = image.new_empty(((2 if self.flip else 1) * n, ch, crop, crop))
            _lib.check(lib.lseg_op_eval_make_crops(P(cur), P(crops), ch, height, width, crop, stride, h_grids, w_grids, int(self.flip), pad, st))
            geo.append((height, width, ph, pw, h_grids, w_grids, stride, crops.shape[0]))
            stacks.append(crops)
        # ... grouped so that at most `max_stack` crops (and their [n, K, crop, crop] fp32 logits: 138 MB per crop at K = 150, crop 480) are
        # alive at once: consecutive scales share a forward while they fit -- a 4:3 ADE image (36 crops) is still ONE batch -- and a larger
        # image / longer scale list runs as several stacks instead of one unbounded allocation
        max_stack = int(getattr(self, "max_stack", 48))
        groups, cur_g, cur_n = [], [], 0
        for i, g_ in enumerate(geo):
            if cur_g and cur_n + g_[7] > max_stack:
                groups.append(cur_g)
                cur_g, cur_n = [], 0
            cur_g.append(i)
            cur_n += g_[7]
        if cur_g:
            groups.append(cur_g)
        for grp in groups:
            allc = torch.cat([stacks[i] for i in grp], dim=0) if len(grp) > 1 else stacks[grp[0]]
            outs_all = self._module_eval(allc, label_set).contiguous()
            assert outs_all.shape == (allc.shape[0], nclass, crop, crop) and outs_all.dtype == torch.float32
            o0 = 0
            for i in grp:
                height, width, ph, pw, h_grids, w_grids, stride, nc = geo[i]
                outs = outs_all[o0:o0 + nc]                                   # a contiguous slice: this scale's crops (+ twins)
                o0 += nc
                smap = image.new_empty((nclass, height, width))
                _lib.check(lib.lseg_op_eval_accumulate(P(outs), P(smap), nclass, height, width, ph, pw, crop, stride, h_grids, w_grids,
                                                       int(self.flip), st))
                _lib.check(lib.lseg_op_eval_resize(P(smap), P(scores), nclass, height, width, h, w, 1, st))
            del outs_all, allc
            for i in grp:
                stacks[i] = None
        return scores

    @torch.no_grad()
    def forward(self, image: torch.Tensor, label_set=None) -> torch.Tensor:
        if image.is_cuda:
            net = getattr(self.module, "net", None)
            if not self.cache_text or net is None or not hasattr(net, "cache_text"):
                return self._forward_device(image, label_set)
            was = net.cache_text
            net.cache_text = True
            try:
                return self._forward_device(image, label_set)
            finally:
                net.cache_text = was
        return self._forward_torch(image, label_set)

    @torch.no_grad()
    def _forward_torch(self, image: torch.Tensor, label_set=None) -> torch.Tensor:
        """The torch statement of the schedule (host tensors; the definition the device kernels are tested against)."""
        batch, _, h, w = image.shape
        assert batch == 1
        nclass = self.nclass if label_set is None else len(label_set)
        crop_size = self.crop_size
        up = self.module._up_kwargs
        scores = image.new_zeros((batch, nclass, h, w))
        for scale in self.scales:
            height, width, gph, gpw, h_grids, w_grids, stride = scale_geometry(h, w, scale, self.base_size, crop_size)
            long_size, short_size = max(height, width), min(height, width)
            cur_img = F.interpolate(image, (height, width), **up)
            if long_size <= crop_size:
                pad_img = _pad_image(cur_img, self.module.mean, self.module.std, crop_size)
                outputs = self._infer(pad_img, label_set)[:, :, :height, :width]
            else:
                pad_img = _pad_image(cur_img, self.module.mean, self.module.std, crop_size) \
                    if short_size < crop_size else cur_img
                _, _, ph, pw = pad_img.shape
                assert (ph, pw) == (gph, gpw)
                boxes, crops = [], []
                for idh in range(h_grids):
                    for idw in range(w_grids):
                        h0, w0 = idh * stride, idw * stride
                        h1, w1 = min(h0 + crop_size, ph), min(w0 + crop_size, pw)
                        boxes.append((h0, h1, w0, w1))
                        crops.append(_pad_image(pad_img[:, :, h0:h1, w0:w1], self.module.mean, self.module.std,
                                                crop_size))
                outs = self._infer(torch.cat(crops, dim=0), label_set)        # one batched pass per scale
                outputs = image.new_zeros((batch, nclass, ph, pw))
                count_norm = image.new_zeros((batch, 1, ph, pw))
                for k, (h0, h1, w0, w1) in enumerate(boxes):                 # same accumulation order
                    outputs[:, :, h0:h1, w0:w1] += outs[k:k + 1, :, :h1 - h0, :w1 - w0]
                    count_norm[:, :, h0:h1, w0:w1] += 1
                assert (count_norm == 0).sum() == 0
                outputs = (outputs / count_norm)[:, :, :height, :width]
            scores += F.interpolate(outputs, (h, w), **up)
        return scores


class SequentialMultiEval(BatchedMultiEval):
    """The reference's own schedule on the same module: every crop and its mirrored twin as separate B = 1 forwards, each re-encoding the
    label set (encoding_models.py:100-131,133-139), torch ops for the data movement.  What `test_lseg.py` costs per image when nothing is
    batched -- the baseline of bench.py's `eval_multiscale` leg and of the crop-480 parity test; never the default."""

    def __init__(self, module, nclass, flip=True, scales=(0.5, 0.75, 1.0, 1.25, 1.5, 1.75)):
        super().__init__(module, nclass, flip=flip, scales=scales, max_batch=1, cache_text=False)

    def _infer(self, crops, label_set):
        ev = (lambda x: self.module.evaluate(x)) if label_set is None else (lambda x: self.module.evaluate_random(x, label_set))
        outs = []
        for i in range(crops.shape[0]):
            x = crops[i:i + 1].contiguous()
            o = ev(x)
            if self.flip:
                o = o + torch.flip(ev(torch.flip(x, dims=[3]).contiguous()), dims=[3])
            outs.append(o)
        return torch.cat(outs, dim=0)

    @torch.no_grad()
    def forward(self, image, label_set=None):
        return self._forward_torch(image, label_set)
