"""Device-side segmentation metrics and cross-entropy value (include/lseg_hip.h: lseg_op_seg_stats).

Replaces the host/numpy step the reference runs on the full [B,K,H,W] logits after every forward:
`batch_pix_accuracy`, `batch_intersection_union` ([3P] encoding/utils/metrics.py, called at
modules/lsegmentation_module.py:49-50,59-60) and the forward value of `SegmentationLosses` =
`nn.CrossEntropyLoss(ignore_index)` ([3P] encoding/nn/loss.py, called at :72).  One pass over the scores on the GPU;
no CPU fallback (CPU tensors raise).
"""
import ctypes as C

import torch

from . import _lib


def seg_stats(scores: torch.Tensor, target: torch.Tensor, ignore_index: int = -1) -> dict:
    """scores fp32 [B,K,H,W] (CUDA), target int64 [B,H,W] (-1 = unlabeled).  Returns python/CPU values:
    correct, labeled (ints), area_inter/area_pred/area_lab/area_union (int64 [K]), nll_sum, nll_count."""
    if not scores.is_cuda:
        raise RuntimeError("lseg_hip.metrics.seg_stats runs on the GPU (no CPU fallback): pass CUDA tensors")
    lib = _lib.load()
    B, K, H, W = scores.shape
    s = scores.detach().float().contiguous()
    t = target.detach().to(scores.device, torch.int64).contiguous()
    if tuple(t.shape) != (B, H, W):
        raise ValueError(f"target shape {tuple(t.shape)} != {(B, H, W)}")
    counts = torch.empty(2 + 3 * K, dtype=torch.int64, device=scores.device)
    nll = torch.empty(2, dtype=torch.float64, device=scores.device)
    st = torch.cuda.current_stream(scores.device).cuda_stream
    _lib.check(lib.lseg_op_seg_stats(C.c_void_p(s.data_ptr()), C.c_void_p(t.data_ptr()), B, K, H, W, int(ignore_index),
                                     C.c_void_p(counts.data_ptr()), C.c_void_p(nll.data_ptr()), C.c_void_p(st)))
    c = counts.cpu()
    n = nll.cpu()
    inter, pred, lab = c[2:2 + K], c[2 + K:2 + 2 * K], c[2 + 2 * K:2 + 3 * K]
    return {"correct": int(c[0]), "labeled": int(c[1]), "area_inter": inter, "area_pred": pred, "area_lab": lab,
            "area_union": pred + lab - inter, "nll_sum": float(n[0]), "nll_count": int(n[1])}


def batch_pix_accuracy(output: torch.Tensor, target: torch.Tensor):
    """[3P] encoding.utils.batch_pix_accuracy(output, target) -> (pixel_correct, pixel_labeled)."""
    r = seg_stats(output, target)
    return r["correct"], r["labeled"]


def batch_intersection_union(output: torch.Tensor, target: torch.Tensor, nclass: int):
    """[3P] encoding.utils.batch_intersection_union(output, target, nclass) -> (area_inter, area_union) as numpy int64."""
    if output.shape[1] != nclass:
        raise ValueError(f"output has {output.shape[1]} classes, nclass={nclass}")
    r = seg_stats(output, target)
    return r["area_inter"].numpy(), r["area_union"].numpy()


def cross_entropy(output: torch.Tensor, target: torch.Tensor, ignore_index: int = -1) -> float:
    """Forward value of nn.CrossEntropyLoss(ignore_index=ignore_index)(output, target) (mean over valid pixels)."""
    r = seg_stats(output, target, ignore_index)
    return r["nll_sum"] / r["nll_count"] if r["nll_count"] else float("nan")


class SegmentationMetric:
    """Device-side stand-in for `encoding.utils.SegmentationMetric(nclass)` as test_lseg.py uses it (:319, 385-388):
    `update(labels, preds)`, `get() -> (pixAcc, mIoU)`, `get_all() -> (pixAcc, mIoU, total_inter, total_union)`, `reset()`.
    `preds` are score tensors [B,K,H,W] (or a list of them, like the evaluator's per-image outputs), `labels` the int64
    masks; the per-image statistics come from ONE pass over the scores on the GPU (lseg_op_seg_stats) instead of an
    arg-max + numpy histograms per image on the host.  `all_reduce()` sums the counters over data-parallel ranks."""

    def __init__(self, nclass: int):
        self.nclass = nclass
        self.reset()

    def reset(self):
        self.total_inter = torch.zeros(self.nclass, dtype=torch.int64)
        self.total_union = torch.zeros(self.nclass, dtype=torch.int64)
        self.total_correct = 0
        self.total_label = 0

    def _accumulate(self, correct: int, labeled: int, inter: torch.Tensor, union: torch.Tensor):
        self.total_correct += int(correct)
        self.total_label += int(labeled)
        self.total_inter += inter.to(torch.int64).cpu()
        self.total_union += union.to(torch.int64).cpu()

    def update(self, labels, preds):
        if torch.is_tensor(preds):
            labels, preds = [labels], [preds]
        for label, pred in zip(labels, preds):
            if pred.dim() == 3:
                pred, label = pred.unsqueeze(0), label.unsqueeze(0)
            r = seg_stats(pred, label)
            self._accumulate(r["correct"], r["labeled"], r["area_inter"], r["area_union"])

    def all_reduce(self):
        from . import dist as D
        t = torch.cat([torch.tensor([self.total_correct, self.total_label], dtype=torch.int64), self.total_inter, self.total_union])
        t = D.sum_over_ranks(t.cuda() if torch.cuda.is_available() and torch.distributed.is_initialized()
                             and torch.distributed.get_backend() == "nccl" else t).cpu()
        self.total_correct, self.total_label = int(t[0]), int(t[1])
        self.total_inter, self.total_union = t[2:2 + self.nclass].clone(), t[2 + self.nclass:].clone()

    def get_all(self):
        eps = 2.220446049250313e-16                      # np.spacing(1), as in [3P] encoding/utils/metrics.py
        pix_acc = 1.0 * self.total_correct / (eps + self.total_label)
        iou = 1.0 * self.total_inter.double() / (eps + self.total_union.double())
        return pix_acc, float(iou.mean()), self.total_inter.numpy(), self.total_union.numpy()

    def get(self):
        pix_acc, miou, _, _ = self.get_all()
        return pix_acc, miou
