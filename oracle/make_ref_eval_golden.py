"""Golden vectors for the multi-scale / flip sliding-window evaluator from the REFERENCE'S OWN CODE
(tests/golden/ref_eval_*.pt).  TEST INFRASTRUCTURE; build container only (needs /root/reference).

Executes additional_utils/encoding_models.py::MultiEvalModule.forward (:54-131) and
additional_utils/models.py::LSeg_MultiEvalModule.forward (:55-132) on CPU around oracle/toy_eval_module.ToyModule.
The reference hard-codes `.cuda()` on freshly created CPU-side buffers; this harness makes `Tensor.cuda` the identity
for the duration of the run -- the only intervention, no reference source is modified.

    python oracle/make_ref_eval_golden.py
"""
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.toy_eval_module import ToyModule                                     # noqa: E402

REF_UTILS = "/root/reference/additional_utils"

# name -> (nclass, base_size, crop_size, image (h, w), scales, flip, labelset size or 0, seed)
EVAL_CASES = {
    "ref_eval_landscape": (5, 40, 32, (37, 52), (0.5, 0.75, 1.0, 1.5), True, 0, 0),
    "ref_eval_portrait_noflip": (4, 48, 32, (61, 40), (0.75, 1.25, 1.75), False, 0, 1),
    "ref_eval_labelset": (6, 40, 32, (45, 45), (0.5, 1.0, 1.25), True, 3, 2),
}


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def run_reference(spec):
    nclass, base, crop, (h, w), scales, flip, nlab, seed = spec
    toy = ToyModule(nclass, base, crop, seed)
    g = torch.Generator().manual_seed(100 + seed)
    image = (torch.rand((1, 3, h, w), generator=g) - 0.5) / 0.5
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        if nlab:
            m = _load(os.path.join(REF_UTILS, "models.py"), "ref_eval_models")
            ev = m.LSeg_MultiEvalModule(toy, device_ids=[], flip=flip, scales=list(scales))
            labelset = [f"l{i}" for i in range(nlab)]
            with torch.no_grad():
                out = ev.forward(image, labelset)
        else:
            m = _load(os.path.join(REF_UTILS, "encoding_models.py"), "ref_eval_encoding_models")
            ev = m.MultiEvalModule(toy, nclass, device_ids=[], flip=flip, scales=list(scales))
            with torch.no_grad():
                out = ev.forward(image)
    finally:
        torch.Tensor.cuda = orig_cuda
    return image, out


def main():
    gd = os.path.join(ROOT, "tests", "golden")
    for name, spec in EVAL_CASES.items():
        image, out = run_reference(spec)
        torch.save({"spec": spec, "image": image.clone(), "scores": out.clone()}, os.path.join(gd, name + ".pt"))
        print(name, tuple(out.shape), float(out.abs().mean()))


if __name__ == "__main__":
    main()
