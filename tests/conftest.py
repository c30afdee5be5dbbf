import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "lang-seg_amd")
for p in (ROOT, SRC):
    if p not in sys.path:
        sys.path.insert(0, p)


# The parity suite runs the training step with DETERMINISTIC reductions (lseg_config.flags bit 3: no fp32 atomics on the gradient path):
# every comparison is reproducible to the bit from run to run, so a tolerance is a property of the seed and not of the launch order
# (VERDICT r4 item 1).  It is also the engine's DEFAULT (no measurable cost); set here explicitly so that the suite does not depend on the
# environment it is started from.  Engines built by the tests -- and by the interpreters they spawn -- read it unless a test passes
# deterministic= explicitly (tests/test_gpu_train.py::test_atomic_sums_stay_within_rounding_of_the_deterministic_ones keeps the default
# atomics covered).
os.environ.setdefault("LSEG_DETERMINISTIC", "1")
# The suite runs on SYNTHETIC weights: hash-based stand-in token ids are fine here, and only here (lseg_hip/tokenizer.py raises without
# this opt-in when neither `clip` nor the CLIP vocabulary is installed -- a real checkpoint must never be scored with them).
os.environ.setdefault("LSEG_SYNTHETIC_TOKENS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    config.addinivalue_line("markers", "gpu_fast: the ~4 minute subset of the gpu tests for iteration -- every operator test, one reference-run "
                                       "fixture per network family, one small training fixture (`-m gpu_fast`; the full `-m gpu` suite runs once per "
                                       "final library)")


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """`-m gpu` on a box without a HIP device: skip (with the reason) instead of 100+ errors.  On a GPU box nothing is
    skipped -- and the product path itself still fails loudly when liblseg_hip.so is missing (lseg_hip/_lib.py)."""
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no HIP device visible (GPU tests run through gpurun on an MI355X)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
