"""lseg_hip.evaluator.BatchedMultiEval against fixtures produced by the REFERENCE'S OWN evaluator code
(additional_utils/encoding_models.py MultiEvalModule.forward, additional_utils/models.py LSeg_MultiEvalModule.forward,
run by oracle/make_ref_eval_golden.py around a deterministic toy network).  The evaluator is host-side data movement
(resize / pad / crop / flip / accumulate / count-normalise), so this runs on CPU."""
import os

import pytest
import torch

from lseg_hip.evaluator import BatchedMultiEval
from oracle.toy_eval_module import ToyModule

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(f[:-3] for f in os.listdir(GOLD) if f.startswith("ref_eval_"))


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("max_batch", [1, 3, 16])
def test_batched_evaluator_equals_the_reference_evaluator(name, max_batch):
    g = torch.load(os.path.join(GOLD, name + ".pt"))
    nclass, base, crop, (h, w), scales, flip, nlab, seed = g["spec"]
    toy = ToyModule(nclass, base, crop, seed)
    ev = BatchedMultiEval(toy, nclass, flip=flip, scales=scales, max_batch=max_batch)
    labelset = [f"l{i}" for i in range(nlab)] if nlab else None
    out = ev.forward(g["image"], labelset)
    assert out.shape == g["scores"].shape
    # same arithmetic per pixel; only the conv's batch blocking may differ in the last bits
    assert (out - g["scores"]).abs().max().item() <= 1e-5 * g["scores"].abs().max().item()
    assert torch.equal(out.argmax(1), g["scores"].argmax(1))
