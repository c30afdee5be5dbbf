#!/usr/bin/env python
"""Where the ViT block's four GEMMs lose time against their own K-loop, on random operands, through the C ABI
(lseg_op_gemm_vit: the engine's specialised epilogues standalone).

  table    one (variant, kind) timing per line.  Run once per build -- LSEG_HIP_LIB picks the attribution build
           (lang-seg_amd/lseg_hip/probe/liblseg_hip_abl<n>.so, `make -C lang-seg_amd/csrc probes`): abl1 = no epilogue stores,
           abl2 = no epilogue at all (the K-loop alone), abl3 = EPI_RES32 without the residual loads, abl4 = EPI_QKV16 writing V like K.
  partial  the same kernels on PART of the chip (max_grid 256 / 128 / 64 / 32 workgroups, rows scaled so every workgroup keeps its tile
           count): if the epilogue share shrinks with the number of CUs bursting at once, the burst is bandwidth-bound.
  stagger  P concurrent launches of 256 / P workgroups each on P streams, started in phase or staggered by 1 / P of a tile time: what
           de-phasing the epilogue bursts across the chip could buy, measured without touching the kernel.
"""
import ctypes as C, json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd"))
import torch
from lseg_hip import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
VARIANT = os.environ.get("LSEG_PROBE_VARIANT", "full")
DT = os.environ.get("PROBE_DTYPE", "fp16")
TD, CD = (torch.float16, _lib.LSEG_F16) if DT == "fp16" else (torch.bfloat16, _lib.LSEG_BF16)
NTOK, NPAD, D = 901, 1024, 1024
KINDS = {"qkv": (3, 3 * D, D), "fc1": (1, 4 * D, D), "proj": (2, D, D), "fc2": (2, D, 4 * D)}      # kind code, N, K


class Problem:
    def __init__(self, kind, M, seed=0):
        self.kind, self.M = kind, M
        self.code, self.N, self.K = KINDS[kind]
        g = torch.Generator(device="cuda").manual_seed(seed)
        self.A = torch.randn((M, self.K), generator=g, device="cuda").to(TD)
        self.W = (torch.randn((self.N, self.K), generator=g, device="cuda") * 0.02).to(TD)
        self.bias = torch.randn((self.N,), generator=g, device="cuda") * 0.1
        self.ntok = NTOK if M % NTOK == 0 else M         # M = B * ntok for the qkv map
        self.npad = (self.ntok + 127) // 128 * 128
        B = M // self.ntok
        if self.code == 3:
            self.q = torch.zeros((B * 16 * self.npad * 64,), dtype=TD, device="cuda")
            self.k = torch.zeros_like(self.q)
            self.v = torch.zeros_like(self.q)
        elif self.code == 2:
            self.q = torch.randn((M, self.N), generator=g, device="cuda")
            self.k = self.v = None
        else:
            self.q = torch.empty((M, self.N), dtype=TD, device="cuda")
            self.k = self.v = None
        self.flops = 2.0 * M * self.N * self.K

    def launch(self, stream, max_grid=0):
        _lib.check(lib.lseg_op_gemm_vit(P(self.A), P(self.W), P(self.bias), P(self.q), P(self.k), P(self.v), self.M, self.N, self.K, CD,
                                        self.code, self.ntok, self.npad, max_grid, C.c_void_p(stream.cuda_stream)))


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3          # us


def table(batch=36):
    st = torch.cuda.current_stream()
    for kind in KINDS:
        pr = Problem(kind, batch * NTOK)
        us = timed(lambda: pr.launch(st))
        print(json.dumps({"mode": "table", "variant": VARIANT, "dtype": DT, "kind": kind, "M": pr.M, "us": round(us, 2),
                          "TF": round(pr.flops / us / 1e6, 1)}), flush=True)
        del pr


def partial():
    st = torch.cuda.current_stream()
    for kind in KINDS:
        for grid in (256, 128, 64, 32):
            pr = Problem(kind, 128 * grid)
            us = timed(lambda: pr.launch(st, grid), iters=12)
            print(json.dumps({"mode": "partial", "variant": VARIANT, "dtype": DT, "kind": kind, "grid": grid, "M": pr.M, "us": round(us, 2),
                              "TF_per_cu": round(pr.flops / us / 1e6 / grid, 3)}), flush=True)
            del pr


def stagger(rounds=3):
    """P concurrent launches (P streams, 256 / P workgroups each, every workgroup keeps `rounds` x the bench's tile count), in phase or
    staggered by d / P of the measured single-tile time; the wall time of the group minus the largest start delay is what a de-phased
    schedule of the same total work would take."""
    main = torch.cuda.current_stream()
    # calibrate torch.cuda._sleep (spin kernel, cycles of the device timer)
    cyc = 2_000_000
    us_per_cycle = timed(lambda: torch.cuda._sleep(cyc), iters=5, warm=1) / cyc
    for kind in KINDS:
        code, N, K = KINDS[kind]
        tiles_per_wg = (128 * N // 256 // 256) * rounds               # with M = 128 * rounds * grid rows
        base = Problem(kind, 128 * rounds * 256)
        t_full = timed(lambda: base.launch(main), iters=8)
        tile_us = t_full / tiles_per_wg
        print(json.dumps({"mode": "stagger", "variant": VARIANT, "kind": kind, "P": 1, "us": round(t_full, 2), "tile_us": round(tile_us, 2),
                          "TF": round(base.flops / t_full / 1e6, 1)}), flush=True)
        del base
        for Pn in (2, 4):
            grid = 256 // Pn
            probs = [Problem(kind, 128 * rounds * grid, seed=i) for i in range(Pn)]
            streams = [torch.cuda.Stream() for _ in range(Pn)]
            for frac in (0.0, 1.0):
                delays = [frac * tile_us * i / Pn for i in range(Pn)]

                def group():
                    ev = torch.cuda.Event()
                    ev.record(main)
                    ends = []
                    for pr, s, d in zip(probs, streams, delays):
                        s.wait_event(ev)
                        with torch.cuda.stream(s):
                            if d > 0:
                                torch.cuda._sleep(int(d / us_per_cycle))
                        pr.launch(s, grid)
                        e = torch.cuda.Event()
                        e.record(s)
                        ends.append(e)
                    for e in ends:
                        main.wait_event(e)
                us = timed(group, iters=8)
                tot = sum(p.flops for p in probs)
                print(json.dumps({"mode": "stagger", "variant": VARIANT, "kind": kind, "P": Pn, "stagger": frac, "us": round(us, 2),
                                  "max_delay_us": round(max(delays), 2), "us_minus_delay": round(us - max(delays), 2),
                                  "TF_net": round(tot / (us - max(delays)) / 1e6, 1)}), flush=True)
            del probs


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "table"
    {"table": table, "partial": partial, "stagger": stagger}[mode]()
