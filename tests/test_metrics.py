"""Metrics / loss: the oracle's restatement of the [3P] encoding helpers against independent implementations
(torch.nn.functional.cross_entropy, a plain loop) on CPU, and the HIP kernel against the oracle on the GPU."""
import numpy as np
import pytest
import torch

from oracle.lseg_oracle import batch_pix_accuracy, batch_intersection_union, cross_entropy_value


def _case(B, K, H, W, seed, unlabeled=0.2, ties=False):
    g = torch.Generator().manual_seed(seed)
    scores = torch.randn((B, K, H, W), generator=g) * 3
    if ties:                                            # fp16-valued logits tie often: first maximum must win
        scores = (scores * 2).round() / 2
    target = torch.randint(0, K, (B, H, W), generator=g)
    target[torch.rand((B, H, W), generator=g) < unlabeled] = -1
    return scores, target


@pytest.mark.parametrize("K", [2, 7, 150])
def test_oracle_metrics_against_independent_formulas(K):
    scores, target = _case(2, K, 24, 20, seed=K, ties=(K == 7))
    correct, labeled = batch_pix_accuracy(scores, target)
    inter, union = batch_intersection_union(scores, target, K)
    pred = scores.argmax(1)
    valid = target >= 0
    assert labeled == int(valid.sum()) and correct == int(((pred == target) & valid).sum())
    for c in range(K):
        i = int(((pred == c) & (target == c)).sum())
        u = int((((pred == c) & valid) | (target == c)).sum())
        assert int(inter[c]) == i and int(union[c]) == u
    ce = cross_entropy_value(scores, target, -1)
    ref = torch.nn.functional.cross_entropy(scores.double(), target, ignore_index=-1)
    assert abs(ce - float(ref)) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 2, 64, 64), (1, 7, 33, 47), (3, 150, 96, 80), (1, 1000, 16, 24)])
def test_hip_seg_stats_matches_oracle(shape):
    from lseg_hip import metrics
    B, K, H, W = shape
    scores, target = _case(B, K, H, W, seed=K + H, ties=(K == 7))
    r = metrics.seg_stats(scores.cuda(), target.cuda())
    correct, labeled = batch_pix_accuracy(scores, target)
    inter, union = batch_intersection_union(scores, target, K)
    assert (r["correct"], r["labeled"]) == (correct, labeled)             # integer counts: bit-exact
    assert torch.equal(r["area_inter"], inter) and torch.equal(r["area_union"], union)
    assert r["nll_count"] == int((target >= 0).sum())
    ce = cross_entropy_value(scores, target, -1)
    assert abs(metrics.cross_entropy(scores.cuda(), target.cuda()) - ce) < 2e-5 * max(1.0, abs(ce))   # fp32 exp/log vs fp64
    a, b = metrics.batch_intersection_union(scores.cuda(), target.cuda(), K)
    assert np.array_equal(a, inter.numpy()) and np.array_equal(b, union.numpy())
    with pytest.raises(RuntimeError):
        metrics.seg_stats(scores, target)                                  # CPU tensors: no fallback


def test_segmentation_metric_accumulation_and_formulas():
    """SegmentationMetric.get(): pixAcc = correct / (eps + labeled), IoU = inter / (eps + union), mIoU = mean over classes
    ([3P] encoding/utils/metrics.py SegmentationMetric.get, used at test_lseg.py:386-388) on oracle counts of two images."""
    from lseg_hip.metrics import SegmentationMetric
    K = 7
    m = SegmentationMetric(K)
    tot_c = tot_l = 0
    ti, tu = torch.zeros(K, dtype=torch.int64), torch.zeros(K, dtype=torch.int64)
    for seed in (1, 2):
        scores, target = _case(1, K, 20, 24, seed=seed)
        c, l = batch_pix_accuracy(scores, target)
        i, u = batch_intersection_union(scores, target, K)
        m._accumulate(c, l, i, u)
        tot_c += c; tot_l += l; ti += i; tu += u
    pix, miou = m.get()
    assert abs(pix - tot_c / tot_l) < 1e-12
    assert abs(miou - float((ti.double() / tu.double().clamp_min(1e-300)).mean())) < 1e-9
    _, _, inter, union = m.get_all()
    assert np.array_equal(inter, ti.numpy()) and np.array_equal(union, tu.numpy())
    m.reset()
    assert m.total_label == 0 and int(m.total_union.sum()) == 0


@pytest.mark.gpu
def test_segmentation_metric_update_on_device():
    from lseg_hip.metrics import SegmentationMetric
    K = 5
    m = SegmentationMetric(K)
    sc, tg = _case(2, K, 16, 16, seed=9)
    m.update(tg.cuda(), sc.cuda())                               # batched tensors
    m.update([tg[0].cuda()], [sc[0].cuda()])                     # list of per-image [K,H,W] scores
    c, l = batch_pix_accuracy(sc, tg)
    c0, l0 = batch_pix_accuracy(sc[:1], tg[:1])
    assert (m.total_correct, m.total_label) == (c + c0, l + l0)
    i, u = batch_intersection_union(sc, tg, K)
    i0, u0 = batch_intersection_union(sc[:1], tg[:1], K)
    assert torch.equal(m.total_inter, i + i0) and torch.equal(m.total_union, u + u0)
