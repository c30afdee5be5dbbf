"""A deterministic stand-in NETWORK for pinning the multi-scale evaluator (TEST INFRASTRUCTURE): the evaluator only moves
data around whatever `module.evaluate` computes, so a seeded 3x3 conv with the LSegModule attribute surface
(modules/lseg_module.py:29-93: base_size, crop_size, mean, std, _up_kwargs, evaluate, evaluate_random) is enough to
compare the reference's MultiEvalModule / LSeg_MultiEvalModule with lseg_hip.evaluator.BatchedMultiEval."""
import torch
import torch.nn.functional as F


class ToyModule(torch.nn.Module):
    def __init__(self, nclass=5, base_size=40, crop_size=32, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.w = torch.nn.Parameter(torch.randn((nclass, 3, 3, 3), generator=g) * 0.3, requires_grad=False)
        self.b = torch.nn.Parameter(torch.randn((nclass,), generator=g), requires_grad=False)
        self.base_size, self.crop_size = base_size, crop_size
        self.mean, self.std = [0.5, 0.5, 0.5], [0.5, 0.5, 0.5]
        self._up_kwargs = {"mode": "bilinear", "align_corners": True}
        self.nclass = nclass

    def evaluate(self, x, target=None):
        assert x.shape[2] == self.crop_size and x.shape[3] == self.crop_size, x.shape   # every crop is padded to crop_size
        col = torch.linspace(0, 1, x.shape[3]).view(1, 1, 1, -1)                          # not flip-symmetric
        return F.conv2d(x, self.w, self.b, padding=1) + 0.25 * col

    def evaluate_random(self, x, labelset, target=None):
        return self.evaluate(x)[:, :len(labelset)]
