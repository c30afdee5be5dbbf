#!/usr/bin/env python
"""What a pure WRITE stream reaches on this chip (the ceiling of upsample4x_planes_scaled_kernel, whose 4.98 GB logits write is 94 % of
its bytes), next to a copy: torch's fill / zero (memset) / copy kernels on a tensor of the logits' size.  GB/s = bytes moved / time."""
import json, sys, torch
n = int(float(sys.argv[1]) if len(sys.argv) > 1 else 36 * 150 * 480 * 480)
x = torch.empty(n, dtype=torch.float32, device="cuda")
y = torch.empty(n, dtype=torch.float32, device="cuda")
def timed(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, fn, nbytes in (("fill_", lambda: x.fill_(1.5), 4 * n), ("zero_", lambda: x.zero_(), 4 * n), ("copy_", lambda: y.copy_(x), 8 * n),
                         ("mul_out", lambda: torch.mul(x, 2.0, out=y), 8 * n)):
    us = timed(fn)
    print(json.dumps({"op": name, "bytes": nbytes, "us": round(us, 1), "GBps": round(nbytes / us / 1e3, 1)}), flush=True)
