#!/usr/bin/env python
"""The one-pass x4 logits upsample (the 4.98 GB write of a B = 36 forward) standalone: time, GB/s on the algorithmic bytes, and bit
equality with the two-stage form.  LSEG_UPS4_VARIANT (read per launch) = 10 * form + (1: plain instead of non-temporal stores) + (100: band
of 8 low rows); forms: 0 direct bilerp per pixel, 1 rolling rows in registers, 2 separable through a third LDS image.
  python tools/upsample_bench.py 36 4 [-- variant ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd"))
import torch
from lseg_hip import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
argv = sys.argv[1:]
variants = [v for v in argv[argv.index("--") + 1:]] if "--" in argv else [os.environ.get("LSEG_UPS4_VARIANT", "0")]
batches = [int(v) for v in (argv[:argv.index("--")] if "--" in argv else argv)] or [36]
for variant in variants:
  os.environ["LSEG_UPS4_VARIANT"] = variant
  for B in batches:
      K, H, W = 150, 120, 120
      g = torch.Generator(device="cuda").manual_seed(0)
      R = torch.zeros((B * K, H + 2, W + 2), device="cuda")
      R[:, 1:-1, 1:-1] = torch.randn((B * K, H, W), generator=g, device="cuda")
      sc = torch.rand((B, 2 * H, 2 * W), generator=g, device="cuda") + 0.5
      out = torch.empty((B, K, 4 * H, 4 * W), device="cuda")
      run = lambda: _lib.check(lib.lseg_op_upsample4x_planes_scaled(P(R), P(sc), P(out), B, K, H, W, 0, None, st))
      for _ in range(3): run()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      torch.cuda.synchronize(); e0.record()
      for _ in range(10): run()
      e1.record(); torch.cuda.synchronize()
      us = e0.elapsed_time(e1) / 10 * 1e3
      nbytes = 4 * (R.numel() + sc.numel() + out.numel())
      same = None
      if B <= 8:
          low = torch.empty((B, K, 2 * H, 2 * W), device="cuda"); ref = torch.empty_like(out)
          _lib.check(lib.lseg_op_upsample4x_planes_scaled(P(R), P(sc), P(ref), B, K, H, W, 1, P(low), st))
          torch.cuda.synchronize()
          same = bool(torch.equal(ref, out))
      print(f"variant={os.environ.get('LSEG_UPS4_VARIANT', 'default')} B={B}: {us:.1f} us -> {nbytes / us / 1e3:.0f} GB/s ({nbytes / us / 1e3 / 8000:.3f} of 8 TB/s)"
            + (f"; equals the two-stage form bit for bit: {same}" if same is not None else ""), flush=True)
