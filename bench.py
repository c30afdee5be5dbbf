#!/usr/bin/env python
"""bench.py -- images/sec of the MI355X-native LSeg forward (BASELINE.json metric).

Workload (config.workload): BASELINE.json configs[1] -- ViT-L/16 + DPT head + CLIP text tower,
480x480, K=150 ADE20K labels, bf16 MFMA inference -- `--batch` images per GPU per step
(default 36), synthetic seeded weights and images (no network: no checkpoints/datasets).
One "step" = one LSegNet.forward call on one batch, INCLUDING the CLIP text tower, which the
reference re-runs on every forward (modules/models/lseg_net.py:183).

MFMA operand type (`dtype`): the reference's image tower is fp32; both 16-bit operand types run at the same MFMA rate, but they
are not equally close to the reference.  With `--dtype auto` (default) the bench measures BOTH -- argmax-mask mismatch fraction and
max |dlogit| against the reference-run fixture tests/golden/ref_full_vitl16_480x480_k150.pt (made by /root/reference's own
LSegNet.forward, oracle/make_ref_golden.py --full), and images/sec on the timed workload -- and times the headline on the one
that meets <= 0.3 % mask flips at equal speed (>= 97 % of the faster one: the box-to-box spread of this bench is +-3 %, and fp16
operands toggle more multiplier bits than bf16, which the chip's power management turns into ~2 % lower clocks on random data);
both results are printed under `dtype_selection`.

Launch:  python bench.py --gpus 1 --steps K --warmup W
   or:   python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
             --master-port P bench.py --gpus N --steps K --warmup W
Multi-GPU = config 3: the image batch is sharded across ranks, weights replicated, NO data-path
collective (weak scaling: per-GPU batch fixed); only the timing barrier/all-reduce(max) uses RCCL.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

# synthetic weights (data: "synthetic"): stand-in token ids are legitimate HERE and nowhere near a real checkpoint (lseg_hip/tokenizer.py)
os.environ.setdefault("LSEG_SYNTHETIC_TOKENS", "1")

GF_IMAGE = lambda K: 799.4 + 0.05898 * K        # image tower GF / image
GF_TEXT = lambda K: 5.959 * K                   # CLIP text tower GF / forward call, the reference's 77-position schedule
# executed by the engine: refinenet1.out_conv (7.55 GF) and head1 (15.10 GF) at 240x240 are replaced by ONE combined 1x1 conv on the
# padded 122x122 map below the upsample (3.90 GF), and the pixel x text correlation runs on that map too (0.01524 GF per label
# instead of 0.05898 at 240x240): DESIGN.md §3.4
GF_IMAGE_EXEC = lambda K: GF_IMAGE(K) - 7.55 - 15.10 + 3.90 - (0.05898 - 0.01524) * K
PEAK_BF16_TFLOPS = 2500.0                       # MI355X_MICROARCH.md dense bf16 MFMA peak


def gf_text_executed(K, L, W=512, layers=12, out_c=512):
    """FLOPs the engine actually spends in the text tower: the exact causal truncation computes L = max(EOT)+1 positions per
    label instead of 77; per layer 24*L*W^2 (qkv, out, mlp) + 4*L^2*W (QK^T, PV), plus the projection."""
    return K * (layers * (24.0 * L * W * W + 4.0 * L * L * W) + 2.0 * W * out_c) / 1e9


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=36,
                    help="images per GPU per step (36 x 901 tokens = 126.7 row blocks of 256: the 256x256 GEMM tiles "
                         "fill 99 %% of their last round; at 32 it is 88 %%)")
    ap.add_argument("--labels", type=int, default=150)
    ap.add_argument("--backbone", default="clip_vitl16_384")
    ap.add_argument("--size", type=int, default=480)
    ap.add_argument("--dtype", default="auto", choices=["auto", "bf16", "fp16"],
                    help="MFMA operand type of the image tower; auto = measure both, run the headline on the one that meets <= 0.3 %% "
                         "argmax flips vs the reference-run fixture at equal speed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-samples", type=int, default=3, help="timed CPU-baseline forwards (each ~5 s)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the batch sweep, the K=1000 leg and the training-step leg")
    ap.add_argument("--no-parity", action="store_true", help="skip the reference-fixture parity leg (dtype auto then means bf16)")
    ap.add_argument("--train-batch", type=int, default=8, help="per-GPU batch of the training-step leg (BASELINE configs[3])")
    ap.add_argument("--train-sync-bn", action="store_true",
                    help="multi-GPU training leg with SyncBatchNorm (the reference's utils.py:34: 56 tiny all-reduces per step); default "
                         "for N > 1 is per-GPU statistics = no collective besides the gradient all-reduce (BASELINE north_star)")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="threads for the CPU baseline (0 = 32 and 64 threads, the faster is reported)")
    ap.add_argument("--no-pmc-traffic", action="store_true",
                    help="do not collect FETCH_SIZE / WRITE_SIZE of the dominant kernel in this run (two rocprofv3 --pmc child passes over 2 "
                         "forwards each, N = 1 only, ~1 min, outside the timed region); roofline.traffic then replays profiles/rNN_traffic.json "
                         "and says so")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)     # the workload of one counter pass: 1 + 2 forwards
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous check without a GPU (tests): every rank joins a gloo group, rank 0 prints {n_gpus, ranks} and leaves")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: re-exec under torch.distributed.run with N ranks on this
    node (one per GPU, 127.0.0.1 rendezvous on a free port) -- a plain invocation must never time ONE GPU and call it N
    (the reference launches its N replicas itself too: utils.py:20-22 DDP, additional_utils/models.py:229-238 threads).
    Fails loudly when fewer than N devices are visible.  Returns the child's exit code."""
    import socket
    import subprocess
    if not args.launch_check:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this node -- refusing to measure fewer devices than asked for")
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on these hosts (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def pmc_child(args):
    """What one rocprofv3 --pmc pass profiles: the headline engine, one warm-up and two forwards of the bench batch."""
    from lseg_hip.config import get_config
    from lseg_hip.engine import HipEngine
    from lseg_hip.synth import synthetic_state_dict, synthetic_tokens, synthetic_images, read_labels
    torch.cuda.set_device(0)
    cfg = get_config(args.backbone)
    labels = read_labels(os.path.join(ROOT, "lang-seg_amd", "label_files", "ade20k_objectInfo150.txt"))[: args.labels]
    eng = HipEngine(cfg, args.size, args.size, max_batch=args.batch, max_labels=len(labels), image_dtype=args.dtype)
    eng.load_state_dict(synthetic_state_dict(cfg, seed=0))
    eng.set_tokens(synthetic_tokens(labels, cfg.text.vocab, cfg.text.ctx))
    x = synthetic_images(args.batch, args.size, args.size, seed=0).cuda()
    for _ in range(3):
        eng.forward(x)
    torch.cuda.synchronize()
    eng.close()


def pmc_symbol_predicate(dom, dtype):
    """Kernel-name test for the dominant symbol's bench-batch instance (256 x 256 tiles; the text tower and the B = 1 self-check
    launch other instantiations)."""
    T = "lseg::F16" if dtype == "fp16" else "lseg::BF16"
    key = dom.split(" ")[0]
    if key == "attention":
        return lambda n: "lseg_attention_kernel<" + T + ", 4, 2, true>" in n
    epi, tag = {"gemm_res32": (3, 0), "gemm_fc1_gelu": (2, 1), "gemm_qkv": (4, 0)}[key]
    return lambda n: ("lseg_gemm_kernel<" + T in n and "TileCfg<256, 256" in n and
                      n.rstrip().endswith(f", false, false, {epi}, {tag}>(lseg::GemmArgs)"))


def pmc_traffic_in_run(dom, dtype, args, alg_bytes):
    """HBM-side traffic of the dominant kernel symbol measured IN THIS RUN: two rocprofv3 child passes (counters only + the kernel-trace
    domain, FETCH_SIZE and WRITE_SIZE separately as MI355X_MICROARCH.md's HBM section prescribes: they do not fit one pass), each over
    `pmc_child` (1 + 2 forwards of the bench batch).  FETCH_SIZE is doubled (gfx950 tallies the 128-byte requests of wide coalesced
    reads at 64 bytes); the figure is L2->fabric bytes, Infinity-Cache hits included.  Returns (dict | None, note)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    pred = pmc_symbol_predicate(dom, dtype)
    tmp = tempfile.mkdtemp(prefix="lseg_pmc_", dir="/tmp")
    per = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = ["rocprofv3", "--pmc", ctr, "--kernel-trace", "-d", d, "-o", "pmc", "--output-format", "csv", "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--dtype", dtype, "--batch", str(args.batch),
                   "--labels", str(args.labels), "--size", str(args.size), "--backbone", args.backbone]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"}, timeout=200,
                                   stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            except subprocess.TimeoutExpired:
                return None, f"rocprofv3 --pmc {ctr} pass timed out"
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {ctr} pass failed (rc {r.returncode}): {r.stderr.decode(errors='replace')[-200:]}"
            disp = {}
            for row in csv.DictReader(open(files[0])):
                if row.get("Counter_Name") == ctr and pred(row["Kernel_Name"]):
                    k_ = row.get("Dispatch_Id") or str(len(disp))
                    disp[k_] = disp.get(k_, 0.0) + float(row["Counter_Value"])
            if not disp:
                return None, f"no dispatch of the dominant symbol in the {ctr} pass"
            per[ctr] = (sum(disp.values()) / len(disp), len(disp))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    b = (2.0 * per["FETCH_SIZE"][0] + per["WRITE_SIZE"][0]) * 1024.0
    return ({"bytes_per_launch": round(b), "algorithmic_bytes": alg_bytes, "over_algorithmic": round(b / alg_bytes, 3) if alg_bytes else None,
             "fetch_bytes_x2": round(2.0 * per["FETCH_SIZE"][0] * 1024.0), "write_bytes": round(per["WRITE_SIZE"][0] * 1024.0),
             "launches_averaged": per["FETCH_SIZE"][1],
             "source": "measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE child passes (separate, counters + kernel-trace "
                       "only) over 2 forwards of the bench batch; FETCH x2 gfx950 correction; L2->fabric bytes incl. Infinity-Cache hits"},
            "ok")


def algorithmic_bytes(dom, M, D, B, npad_tok):
    """Bytes one launch of the dominant symbol must move once (16-bit operands, fp32 residual stream)."""
    key = dom.split(" ")[0]
    if key == "gemm_fc1_gelu":
        return 2 * (M * D + 4 * D * D + M * 4 * D)
    if key == "gemm_res32":         # attn.proj and mlp.fc2 share the symbol: the per-launch average of the two shapes
        return (2 * (M * D + D * D) + 8 * M * D + 2 * (M * 4 * D + 4 * D * D) + 8 * M * D) // 2
    if key == "gemm_qkv":
        return 2 * (M * D + 3 * D * D + M * 3 * D)
    return 2 * 4 * B * (D // 64) * npad_tok * 64


def launch_check(args):
    """--launch-check: the N-rank rendezvous and the reductions the bench line depends on, on gloo, no GPU touched."""
    from lseg_hip import dist as D
    rank, local_rank, world = D.init_from_env("gloo")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    slow = D.max_over_ranks(1.0 + rank)
    rates = D.gather_floats(10.0 * (rank + 1))
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "max_over_ranks": slow, "per_rank": rates}), flush=True)
    D.barrier()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def _cpu_forward_times(cfg, sd, tok, size, threads, samples):
    """`samples` timed B = 1 oracle forwards at `threads` torch threads (after a warm-up on the reduced twin)."""
    from oracle.lseg_oracle import lseg_forward
    from lseg_hip.config import get_config
    from lseg_hip.synth import synthetic_images, synthetic_state_dict, synthetic_tokens
    torch.set_num_threads(threads)
    tiny = get_config("tiny16")
    times = []
    with torch.no_grad():
        lseg_forward(synthetic_state_dict(tiny), synthetic_images(1, 64, 64),
                     synthetic_tokens(["a", "b"], tiny.text.vocab, tiny.text.ctx), tiny)
        x = synthetic_images(1, size, size, seed=0)
        for _ in range(max(1, samples)):
            t0 = time.time()
            lseg_forward(sd, x, tok, cfg)
            times.append(time.time() - t0)
    return times


def cpu_baseline(cfg, sd, tok, size, threads, samples=3, args=None):
    """The CPU oracle (a port of the reference forward, oracle/lseg_oracle.py) timed on this box's host cores.  BOUNDED sample: B = 1
    forwards of the same workload (fp32 image tower + fp16-emulated text tower recomputed, reference semantics); the median is reported.
    SURVEY.md par. 8d asks for os.cpu_count() threads, but torch's CPU GEMMs thrash far below the core count of a two-socket 256-core host:
    two forwards did not finish in 90 s at all 256 cores (round 4) nor in 45 s pinned to one NUMA node's 128 (round 5, lease B), against
    1.9 s at 32 threads.  So: 32 threads (`samples` forwards) and 64 threads (2 forwards), both in this process; `value` is the faster of
    the two with its own `cores`; both are listed under `by_threads`.  ~10-15 s in total."""
    ncpu = os.cpu_count() or 1
    base = threads or min(32, ncpu)
    by = {base: {"seconds": [round(t, 2) for t in _cpu_forward_times(cfg, sd, tok, size, base, samples)]}}
    if not threads and ncpu >= 64:
        by[64] = {"seconds": [round(t, 2) for t in _cpu_forward_times(cfg, sd, tok, size, 64, 2)]}
    med = {n: sorted(v["seconds"])[len(v["seconds"]) // 2] for n, v in by.items() if v.get("seconds")}
    best = min(med, key=med.get)
    for n in med:
        by[n]["images_per_sec"] = round(1.0 / med[n], 4)
    return {"value": round(1.0 / med[best], 4), "unit": "images/sec", "cores": best, "kind": "port",
            "by_threads": {str(n): v for n, v in by.items()}, "host_cores": ncpu,
            "sample": f"median of the timed B=1 forwards of the same workload (torch-CPU oracle, text tower recomputed): {len(by[base]['seconds'])} at "
                      f"{base} threads" + (", 2 at 64 threads" if 64 in by else "") + ", in-process; value = the faster thread count; larger thread "
                      "counts thrash on this host (128 pinned to one NUMA node: > 45 s for 2 forwards; 256: > 90 s) and are not attempted"}


def build_id():
    """The commit the shipped library was built at (written by __graft_entry__.build(); .git does not travel to the GPU box)."""
    try:
        import subprocess
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
    except Exception:                                     # noqa: BLE001
        pass
    try:
        return open(os.path.join(ROOT, "lang-seg_amd", "lseg_hip", "_build_id.txt")).read().strip()
    except OSError:
        return None


def time_forward(eng, x, steps, warmup, sync):
    for _ in range(warmup):
        eng.forward(x)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.forward(x)
    sync()
    return (time.perf_counter() - t0) / steps


_PARITY_SD = {}


def parity_floor(name):
    """profiles/r06_parity_floor.json (tools/parity_floor.py, CPU): argmax flips between the fp32 oracle and the fp32 reference on the
    same fixture -- what "bit-exact masks" means between two fp32 implementations.  None when the file / fixture entry is missing."""
    try:
        fl = json.load(open(os.path.join(ROOT, "profiles", "r06_parity_floor.json"))).get(name, {})
        return fl.get("oracle_vs_reference_240", {}).get("argmax_mismatch_frac")
    except (OSError, ValueError):
        return None


PARITY_FIXTURES = ("ref_full_vitl16_480x480_k150", "ref_full_vitl16_480x480_k150_outlier")


def parity_vs_reference(dtype, name="ref_full_vitl16_480x480_k150"):
    """Engine (production schedule, B=1) against tests/golden/<name>.pt -- outputs of the REFERENCE'S OWN LSegNet.forward at BASELINE
    configs[1] (oracle/make_ref_golden.py --full; the fixture travels, /root/reference does not): fraction of the 240x240 argmax mask that
    differs from the reference's, max |dlogit| on the stored logits (every 8th pixel of every label plane + the reference's two best labels
    of EVERY pixel) and the largest reference top-2 margin at a differing pixel.  `*_outlier` = the same configuration on
    realistic-statistics weights (residual outlier channels ~1e3, LayerNorm gains 10, large BatchNorm scales: lseg_hip.synth.outlier_state_dict)."""
    from lseg_hip.config import get_config
    from lseg_hip.engine import HipEngine
    from lseg_hip.synth import synthetic_images, fixture_state_dict
    path = os.path.join(ROOT, "tests", "golden", name + ".pt")
    if not os.path.exists(path):
        return None
    g = torch.load(path)
    bb, H, W, B, K, seed, arch, depth = g["spec"]
    cfg = get_config(bb, arch_option=arch, block_depth=depth, activation="lrelu")
    eng = HipEngine(cfg, H, W, max_batch=B, max_labels=K, image_dtype=dtype)
    if name not in _PARITY_SD:
        _PARITY_SD[name] = fixture_state_dict(cfg, seed, g)
    eng.load_state_dict(_PARITY_SD[name])
    eng.set_tokens(g["tokens"])
    eng.forward(synthetic_images(B, H, W, seed=seed).cuda())
    low = eng.intermediate("lowres", (B, K, H // 2, W // 2)).cpu()
    rng = eng.check_range()
    eng.close()
    err = max((low[:, :, ::8, ::8] - g["lowres_sub8"].float()).abs().max().item(),
              (low.gather(1, g["top2_idx"].long()) - g["top2_val"].float()).abs().max().item())
    mism = low.argmax(1) != g["argmax_lowres"].long()
    margin = g["margin_lowres"].float()
    return {"fixture": name + ".pt (reference-run)", "argmax_mismatch_frac": round(mism.float().mean().item(), 6),
            "max_abs_dlogit": round(err, 5), "logit_absmax": round(float(g["lowres_absmax"]), 3),
            "max_reference_margin_at_mismatch": round(margin[mism].max().item() if mism.any() else 0.0, 5),
            "nonfinite_16bit_activations": int(rng["nonfinite"]), "max_abs_16bit_activation": round(float(rng["max_abs"]), 1),
            "fp32_vs_fp32_floor_frac": parity_floor(name),
            "mismatch_over_floor": (round(mism.float().mean().item() / parity_floor(name), 1) if parity_floor(name) else None)}


def boundary_leg(args, cfg, sd, labels, dtype, x, sync):
    """the DROP-IN (LSegNet.forward in eval mode: stamp check, token check, range guard, engine call) timed beside the bare
    HipEngine.forward of the very engine it drives, same inputs, B = 1 / 4 / bench batch; both include the text tower like the reference."""
    import warnings
    from modules.models.lseg_net import LSegNet
    warnings.simplefilter("ignore")
    net = LSegNet(labels=labels, backbone=args.backbone, features=cfg.features, arch_option=0, block_depth=0, activation="lrelu", image_dtype=dtype)
    net.load_state_dict(sd)
    net = net.cuda().eval()
    out = {}
    with torch.no_grad():
        for b in sorted({1, 4, x.shape[0]}):
            xb = x[:b]
            net(xb)                                            # builds / packs the engine for this batch size
            eng = net._last_engine
            n = 40 if b == 1 else (20 if b <= 4 else 8)
            t_net = time_forward(type("F", (), {"forward": staticmethod(lambda v: net(v))}), xb, n, 3, sync)
            t_eng = time_forward(eng, xb, n, 3, sync)
            t_net2 = time_forward(type("F", (), {"forward": staticmethod(lambda v: net(v))}), xb, n, 1, sync)
            t_net = min(t_net, t_net2)
            out[str(b)] = {"lsegnet_forward_images_per_sec": round(b / t_net, 1), "engine_forward_images_per_sec": round(b / t_eng, 1),
                           "boundary_over_engine": round(t_net / t_eng, 4)}
    for e in list(net._engines.values()):
        e.close()
    net._engines.clear()
    del net
    torch.cuda.empty_cache()
    return {"what": "modules.models.lseg_net.LSegNet(...).eval()(x) vs HipEngine.forward on the same engine (text tower recomputed in both); "
                    "ratio > 1 = host-side cost of the drop-in class", "dtype": dtype, "by_batch": out}


def eval_leg(args, cfg, sd, labels, dtype, sync):
    """the workload test_lseg.py runs per image (additional_utils/encoding_models.py:54-139): 6 scales + flip, crop 480 /
    base 520, one 512x683 image, ViT-L/16, K = 150 -- BatchedMultiEval (ALL crops of ALL scales + their mirrored twins in one batch-36 forward, data movement in
    csrc/evaluator.hip, label set encoded once) beside the reference's literal schedule on the same network (36 B = 1 forwards, each
    re-encoding the labels, torch ops for pad / crop / flip / accumulate)."""
    import warnings
    from modules.models.lseg_net import LSegNet
    from lseg_hip.evaluator import BatchedMultiEval, SequentialMultiEval, ModuleSurface, scale_geometry
    from lseg_hip.synth import synthetic_images
    warnings.simplefilter("ignore")
    net = LSegNet(labels=labels, backbone=args.backbone, features=cfg.features, arch_option=0, block_depth=0, activation="lrelu", image_dtype=dtype)
    net.load_state_dict(sd)
    mod = ModuleSurface(net.cuda().eval(), crop_size=480, base_size=520)
    img = synthetic_images(1, 512, 683, seed=9).cuda()
    scales = (0.5, 0.75, 1.0, 1.25, 1.5, 1.75)
    crops = [scale_geometry(512, 683, s_, 520, 480)[4] * scale_geometry(512, 683, s_, 520, 480)[5] for s_ in scales]
    res = {}
    with torch.no_grad():
        for tag, ev, n in (("batched", BatchedMultiEval(mod, len(labels), flip=True, scales=scales), 3),
                           ("sequential_reference_schedule", SequentialMultiEval(mod, len(labels), flip=True, scales=scales), 2)):
            got = ev(img)
            sync()
            t0 = time.perf_counter()
            for _ in range(n):
                got = ev(img)
            sync()
            res[tag] = {"images_per_sec": round(n / (time.perf_counter() - t0), 3), "seconds_per_image": round((time.perf_counter() - t0) / n, 4)}
            res[tag + "_scores"] = got
    a, b = res.pop("batched_scores"), res.pop("sequential_reference_schedule_scores")
    d = (a - b).abs().max().item()
    res["max_abs_score_difference"] = round(d, 5)
    res["argmax_agreement"] = round((a.argmax(1) == b.argmax(1)).float().mean().item(), 6)
    res["speedup"] = round(res["batched"]["images_per_sec"] / res["sequential_reference_schedule"]["images_per_sec"], 2)
    res["forwards_per_image"] = {"crops_per_scale": crops, "reference_b1_forwards": 2 * sum(crops), "batched_forwards": 1}
    res["what"] = ("one 512x683 image, scales 0.5-1.75 + flip, crop 480 / base 520, K = 150, " + dtype + " operands; scores = sum over scales of "
                   "the count-normalised overlap-added crop logits (not bit-equal between the two: split-K at small batches and torch-vs-device "
                   "resize round differently; tests/test_gpu_evaluator.py holds the batch-invariant schedule to 1e-5)")
    for e in list(net._engines.values()):
        e.close()
    net._engines.clear()
    del net, mod
    torch.cuda.empty_cache()
    return res


def train_leg(cfg, sd, tok, size, B, rank, sync, D, sync_bn=True):
    """BASELINE configs[3]: one data-parallel training step per GPU batch B -- train-mode forward, fused CE, backward with the
    bucketed RCCL all-reduce overlapped, fused SGD (lseg_hip/train.py).  1 warm-up + 3 timed steps."""
    from lseg_hip.engine import HipEngine
    from lseg_hip.synth import synthetic_images
    from lseg_hip.train import DataParallelTrainer
    sd_dev = {k: v.cuda() for k, v in sd.items()}
    eng = HipEngine(cfg, size, size, max_batch=B, max_labels=tok.shape[0], image_dtype="bf16")
    eng.load_state_dict(sd_dev)
    eng.set_tokens(tok)
    tr = DataParallelTrainer(eng, sd_dev, sync_bn=sync_bn)
    x = synthetic_images(B, size, size, seed=100 + rank).cuda()
    g = torch.Generator().manual_seed(7 + rank)
    t = torch.randint(0, tok.shape[0], (B, size, size), generator=g)
    t[torch.rand(t.shape, generator=g) < 0.2] = -1
    t = t.cuda()
    base_lr = 0.004 / 16 * B                                    # lsegmentation_module.py:32, train.sh
    loss = tr.step(x, t, base_lr, 10 * base_lr)
    sync()
    t0 = time.perf_counter()
    for _ in range(3):
        loss = tr.step(x, t, base_lr, 10 * base_lr)
    sync()
    dt = D.max_over_ranks((time.perf_counter() - t0) / 3, device="cuda")
    exch = ("bucketed gradient all-reduce (RCCL, launched from the engine's bucket callbacks under the remaining backward)" if tr.world > 1
            else "no collective at N=1 (the bucket callbacks fire, nothing is exchanged)")
    out = {"images_per_sec": round(tr.world * B / dt, 2), "ms_per_step": round(dt * 1e3, 2), "per_gpu_batch": B,
           "loss": round(float(loss.item()), 4), "sync_bn": tr.sync_bn,
           "what": f"train-mode forward + fused x2-upsample/CE + backward + {exch} + fused SGD; bf16 MFMA operands, fp32 masters/gradients; "
                   "synthetic images and masks",
           "tflops_3x_forward_convention": round(tr.world * 3 * B * GF_IMAGE(tok.shape[0]) / dt / 1e3, 1)}
    if tr.world > 1:
        out["exchange"] = exchange_report(tr, x, t, base_lr, sync, D, dt)
    eng.close()
    return out


def exchange_report(tr, x, t, base_lr, sync, D, dt_with):
    """What the first multi-GPU run must show about the gradient exchange, outside the timed steps: the world size as
    the process group reports it, every bucket's all-reduce time on the exchange stream and its bus bandwidth (ring convention:
    2 (N - 1) / N x bytes / time), and the step time with the exchange switched off -- the difference is the EXPOSED communication."""
    import torch.distributed as dist
    ex = tr.exchange
    rep = {"backend": dist.get_backend(ex.group), "world_size_reported_by_group": dist.get_world_size(ex.group),
           "reduce_op": "avg (ncclAvg, no extra launch)" if ex._avg_op else "sum + one multi-tensor divide", "buckets": len(ex)}
    ex.timing = True
    tr.step(x, t, base_lr, 10 * base_lr)
    sync()
    times = ex.bucket_times_ms()
    ex.timing = False
    n = ex.world
    rows = []
    for i in sorted(times):
        mb = ex.buckets[i].numel() * 4 / 1e6
        ms = times[i]
        rows.append({"bucket": i, "MB": round(mb, 1), "ms": round(ms, 3), "bus_GBps": round(2 * (n - 1) / n * mb / max(ms, 1e-6), 1)})
    rep["per_bucket"] = rows
    rep["allreduce_ms_total"] = round(sum(r["ms"] for r in rows), 3)
    # the same step without the collectives (callbacks detached; gradients stay local): exposed communication = with - without
    world, ex.world = ex.world, 1
    try:
        tr.step(x, t, base_lr, 10 * base_lr)
        sync()
        t0 = time.perf_counter()
        for _ in range(3):
            tr.step(x, t, base_lr, 10 * base_lr)
        sync()
        dt_without = D.max_over_ranks((time.perf_counter() - t0) / 3, device="cuda")
    finally:
        ex.world = world
    rep["ms_per_step_exchange_off"] = round(dt_without * 1e3, 2)
    rep["exposed_exchange_ms"] = round((dt_with - dt_without) * 1e3, 2)
    rep["overlap_fraction"] = round(1.0 - max(0.0, dt_with - dt_without) * 1e3 / max(rep["allreduce_ms_total"], 1e-6), 3)
    return rep


def k1000_leg(cfg, sd, size, B, sync, D, dtype):
    """BASELINE configs[4]: K = 1000 open-vocabulary prompts (the FSS-1000 class names), skinny pixel x text correlation stress."""
    from lseg_hip.engine import HipEngine
    from lseg_hip.synth import synthetic_images, synthetic_tokens, read_labels
    names = read_labels(os.path.join(ROOT, "lang-seg_amd", "label_files", "fewshot_fss.txt"), skip_header=False)[:1000]
    tok = synthetic_tokens(names, cfg.text.vocab, cfg.text.ctx)
    eng = HipEngine(cfg, size, size, max_batch=B, max_labels=len(names), image_dtype=dtype)
    eng.load_state_dict(sd)
    eng.set_tokens(tok)
    x = synthetic_images(B, size, size, seed=5).cuda()
    dt = D.max_over_ranks(time_forward(eng, x, 5, 2, sync), device="cuda")
    eng.close()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    par = None
    if int(os.environ.get("RANK", "0")) == 0:
        try:                                   # the configuration is reported WITH its own parity, not under configs[1]'s
            par = parity_vs_reference(dtype, "ref_full_vitl16_480x480_k1000")
            _PARITY_SD.clear()
            if par:
                par["meets_headline_rule_le_0p3pct"] = bool(par["argmax_mismatch_frac"] <= 0.003)
        except Exception as e:                 # noqa: BLE001
            par = {"error": f"{type(e).__name__}: {e}"}
    return {"parity": par, "images_per_sec": round(world * B / dt, 1), "ms_per_step": round(dt * 1e3, 2), "per_gpu_batch": B, "labels": len(names), "dtype": dtype,
            "what": "LSegNet.forward with 1000 prompts, text tower recomputed every call, full [B,1000,480,480] fp32 logits written"}


def latest_traffic_file():
    """profiles/rNN_traffic.json of the newest round (written by tools/collect_profiles.sh + tools/make_traffic_json.py on a GPU box)."""
    pd = os.path.join(ROOT, "profiles")
    cands = sorted(f for f in os.listdir(pd) if f.startswith("r") and f.endswith("_traffic.json")) if os.path.isdir(pd) else []
    return os.path.join(pd, cands[-1]) if cands else None


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))          # N ranks, one per GPU; this process only waits for them
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: the line would mislabel the device count")
    if args.launch_check:
        return launch_check(args)
    if args.pmc_child:
        return pmc_child(args)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the LSeg HIP engine has no CPU fallback")
    from lseg_hip import dist as D
    if torch.cuda.device_count() <= int(os.environ.get("LOCAL_RANK", "0")):
        raise SystemExit(f"rank {os.environ.get('RANK')}: no device {os.environ.get('LOCAL_RANK')} on this node ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    rank, local_rank, world = D.init_from_env("nccl")        # "nccl" == RCCL on ROCm
    dist = None
    if world > 1:
        import torch.distributed as dist

    from lseg_hip.config import get_config
    from lseg_hip.engine import HipEngine
    from lseg_hip.synth import synthetic_state_dict, synthetic_tokens, synthetic_images, read_labels

    cfg = get_config(args.backbone)
    sd = synthetic_state_dict(cfg, seed=0)
    labels = read_labels(os.path.join(ROOT, "lang-seg_amd", "label_files", "ade20k_objectInfo150.txt"))
    if args.labels > len(labels):
        extra = read_labels(os.path.join(ROOT, "lang-seg_amd", "label_files", "fewshot_fss.txt"), skip_header=False)
        labels = (labels + extra)[: args.labels]
    labels = labels[: args.labels]
    tok = synthetic_tokens(labels, cfg.text.vocab, cfg.text.ctx)
    K, B = len(labels), args.batch
    # every rank gets its own shard of the global batch (different seed = different images)
    x = synthetic_images(B, args.size, args.size, seed=rank).cuda()

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- MFMA operand type: parity against the reference-run fixture + a short timing probe of both, BEFORE the timed region -------
    headline = args.backbone == "clip_vitl16_384" and args.size == 480 and K == 150
    cands = ["bf16", "fp16"] if args.dtype == "auto" else [args.dtype]
    parity = {}
    parity_all = {}
    if headline and not args.no_parity:
        for dt_ in ("bf16", "fp16"):
            per = {n_: parity_vs_reference(dt_, n_) for n_ in PARITY_FIXTURES}
            per = {n_: v for n_, v in per.items() if v}
            parity_all[dt_] = per
            # the selection runs on the WORSE of the fixtures (N(0, 0.02)-style weights and realistic-statistics weights): VERDICT r4 item 2
            parity[dt_] = max(per.values(), key=lambda v: v["argmax_mismatch_frac"]) if per else None
        _PARITY_SD.clear()
    engines, probe = {}, {}
    for dt_ in cands:
        e = HipEngine(cfg, args.size, args.size, max_batch=B, max_labels=K, image_dtype=dt_)
        e.load_state_dict(sd)
        e.set_tokens(tok)
        engines[dt_] = e
        if len(cands) > 1:
            probe[dt_] = B / D.max_over_ranks(time_forward(e, x, 5, 2, sync), device="cuda")
    chosen = cands[0]
    rule = "given on the command line"
    if len(cands) > 1:
        ok = {d: parity.get(d) is not None and parity[d]["argmax_mismatch_frac"] <= 0.003 and
              not any(v.get("nonfinite_16bit_activations") for v in parity_all.get(d, {}).values()) for d in cands}
        fast = max(probe.values())
        rule = ("the operand type that meets <= 0.3 % argmax flips vs the reference-run fixtures (the WORSE of the N(0, 0.02)-weights fixture and the "
                "realistic-statistics `_outlier` one, no non-finite 16-bit activation on either) at >= 97 % of the faster one's probe rate; "
                "ties and no-qualifier -> the closer one, then bf16")
        qual = [d for d in cands if ok[d] and probe[d] >= 0.97 * fast]
        if qual:
            chosen = min(qual, key=lambda d: parity[d]["argmax_mismatch_frac"])
        elif all(parity.get(d) for d in cands):
            chosen = min((d for d in cands if probe[d] >= 0.97 * fast), key=lambda d: parity[d]["argmax_mismatch_frac"])
        if dist is not None:                     # one decision for the whole job: rank 0's
            box = [chosen]
            dist.broadcast_object_list(box, src=0)
            chosen = box[0]
    eng = engines[chosen]

    # ---- which kernel symbol dominates: one untimed pass with every family bracketed by HIP events ---------------------------------
    fams = ["mlp_fc1", "mlp_fc2", "attn_proj", "attn_qkv", "attention", "layernorm", "correlation"]
    eng.set_profiling(["forward"] + fams)
    for _ in range(2):
        eng.forward(x)
    torch.cuda.synchronize()
    fam = {f: eng.profile(f) for f in ["forward"] + fams}
    eng.set_profiling(False)
    # kernel symbols: proj and fc2 are the same instance (lseg_gemm_kernel<.., EPI_RES32>); the others have one family each
    symbols = {"gemm_res32 (attn.proj + mlp.fc2, fp32 residual epilogue)": ["attn_proj", "mlp_fc2"], "gemm_fc1_gelu (mlp.fc1 + bias + GELU)": ["mlp_fc1"],
               "gemm_qkv (attn.qkv, head-major q/k, transposed v)": ["attn_qkv"], "attention (fused QK^T softmax PV)": ["attention"]}
    sym_ms = {s_: sum(fam[f]["total_ms"] for f in fs) for s_, fs in symbols.items()}
    dom = max(sym_ms, key=sym_ms.get)
    dom_fams = symbols[dom]

    for _ in range(args.warmup):
        out = eng.forward(x)
    eng.set_profiling(["forward"] + dom_fams)   # timing events are created here, outside the timed region
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = eng.forward(x)
    sync()
    dt = time.perf_counter() - t0
    dt_local = dt
    prof = {f: eng.profile(f) for f in ["forward"] + dom_fams}
    eng.set_profiling(False)
    dt = D.max_over_ranks(dt, device="cuda")
    assert torch.isfinite(out).all(), "non-finite logits"
    # Self-check outside the timed region: the tile configuration follows the problem size, so the batch the bench
    # times is checked against a single-image run of the same engine -- the configuration the oracle-parity tests
    # validate (tests/test_gpu_forward.py).  A mismatch voids the number.
    h2 = args.size // 2
    low_b = eng.intermediate("lowres", (B, K, h2, h2))[0].clone()
    eng.forward(x[:1])
    low_1 = eng.intermediate("lowres", (1, K, h2, h2))[0]
    selfcheck = (low_b - low_1).abs().max().item()
    if not selfcheck <= 1e-2:
        raise SystemExit(f"bench self-check failed: batch-of-{B} logits differ from the single-image run by {selfcheck}")
    # the other operand type on the same workload (outside the headline's timed region)
    other = {}
    for dt_, e in engines.items():
        if dt_ != chosen:
            other[dt_] = B * world / D.max_over_ranks(time_forward(e, x, min(args.steps, 10), 2, sync), device="cuda")

    # HBM-side traffic of the dominant symbol, collected in this run by two counter-only child passes (N = 1; outside the timed region)
    pmc_now, pmc_note = None, "disabled (--no-pmc-traffic)" if args.no_pmc_traffic else "N > 1: not collected"
    if rank == 0 and world == 1 and not args.no_pmc_traffic:
        torch.cuda.synchronize()
        Mtok_ = B * cfg.tokens(args.size, args.size)
        npad = (cfg.tokens(args.size, args.size) + 127) // 128 * 128
        try:
            pmc_now, pmc_note = pmc_traffic_in_run(dom, chosen, args, algorithmic_bytes(dom, Mtok_, cfg.dim, B, npad))
        except Exception as e:                                   # noqa: BLE001  (the line must survive a failing profiler)
            pmc_now, pmc_note = None, f"{type(e).__name__}: {e}"

    per_rank = D.gather_floats(B * args.steps / dt_local, device="cuda")
    if rank == 0:
        ms = dt / args.steps * 1e3
        ips = world * B * args.steps / dt
        # ---- roofline of the dominant kernel symbol, measured with HIP events inside the timed region -----------------------------
        n_l = sum(prof[f]["launches"] for f in dom_fams)
        t_ms = sum(prof[f]["total_ms"] for f in dom_fams)
        fl = sum(prof[f]["flops_per_launch"] * prof[f]["launches"] for f in dom_fams)
        roof = None
        if n_l:
            ach = fl / (t_ms * 1e-3) / 1e12
            traffic = pmc_now                       # measured in this run (below the timed region), or the replayed profile
            tpath = latest_traffic_file()
            if traffic is None and tpath:
                tj = json.load(open(tpath))
                key = {"gemm_res32": "res32_gemm", "gemm_fc1_gelu": "mlp_fc1_gemm", "gemm_qkv": "qkv_gemm", "attention": "attention"}[dom.split(" ")[0]]
                t = tj.get(key, {})
                if t.get("batch", B) == B and "traffic_bytes_per_launch" in t and t.get("dtype", "bf16") == chosen:
                    traffic = {"bytes_per_launch": round(t["traffic_bytes_per_launch"]), "algorithmic_bytes": t.get("algorithmic_bytes_per_launch"),
                               "replayed_from": os.path.relpath(tpath, ROOT), "collected_at_commit": tj.get("_meta", {}).get("commit"),
                               "in_run_collection": pmc_note,
                               "source": "NOT measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/collect_profiles.sh), "
                                         "FETCH x2 gfx950 correction; L2->fabric bytes incl. Infinity-Cache hits"}
            roof = {"bound": "mfma", "kernel": f"lseg_gemm_kernel / lseg_attention_kernel family: {dom}; {chosen} operands; tile by problem size",
                    "families": dom_fams, "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                    "avg_launch_ms": round(t_ms / n_l, 5), "launches": n_l, "flops_per_launch": fl / n_l,
                    "share_of_forward": round(t_ms / max(prof["forward"]["total_ms"], 1e-9), 4)}
        # every ViT-block kernel family from the untimed all-families pass (2 forwards): TFLOP/s vs the MFMA peak, LayerNorm vs HBM
        Mtok = B * cfg.tokens(args.size, args.size)
        kern = {}
        for f in fams:
            p_ = fam[f]
            if not p_["launches"]:
                continue
            avg = p_["total_ms"] / p_["launches"]
            if f == "correlation":
                # csrc/corr.hip: label planes + cell dot products in one pass; the family's "flops" slot carries its ALGORITHMIC BYTES
                # (g read once, planes + gram records written, T once: DESIGN par. 3.4)
                gb = p_["flops_per_launch"] / 1e9
                kern[f] = {"bound": "hbm", "kernel": "corr_planes_kernel (pixel x text label planes + 2x2-cell dot products of g, one pass)",
                           "avg_launch_ms": round(avg, 5), "algorithmic_MB": round(gb * 1e3, 1), "achieved_GBps": round(gb / (avg * 1e-3), 1),
                           "frac": round(gb / (avg * 1e-3) / 8000.0, 4),
                           "mfma_TFLOPs": round(2.0 * K * 512 * B * (args.size // 4 + 2) ** 2 / (avg * 1e-3) / 1e12, 1)}
            elif f == "layernorm":
                gb = Mtok * cfg.dim * (4 + 2) / 1e9          # fp32 row in, 16-bit row out
                kern[f] = {"bound": "hbm", "avg_launch_ms": round(avg, 5), "achieved_GBps": round(gb / (avg * 1e-3), 1),
                           "frac": round(gb / (avg * 1e-3) / 8000.0, 4)}
            else:
                tf = p_["flops_per_launch"] / (avg * 1e-3) / 1e12
                kern[f] = {"bound": "mfma", "avg_launch_ms": round(avg, 5), "achieved_TFLOPs": round(tf, 1), "frac": round(tf / PEAK_BF16_TFLOPS, 4)}
        # whole-path figures.  Two conventions, both labelled: EXECUTED = the FLOPs the engine spends (text tower
        # truncated to max(EOT)+1 positions: exact); REFERENCE-ALGORITHM = what the reference's schedule would spend
        # on the same inputs (77 text positions).  Roofline fractions use the executed count only.
        L_exec = int(tok.argmax(dim=-1).max().item()) + 1
        gf_step = B * GF_IMAGE_EXEC(K) + gf_text_executed(K, L_exec)
        gf_step_ref = B * GF_IMAGE(K) + GF_TEXT(K)
        sel = {"rule": rule, "chosen": chosen}
        for d in ("bf16", "fp16"):
            e_ = {}
            if d == chosen:
                e_["images_per_sec"] = round(ips, 1)
            elif d in other:
                e_["images_per_sec"] = round(other[d], 1)
            if d in probe:
                e_["probe_images_per_sec_per_gpu"] = round(probe[d], 1)
            if parity.get(d):
                e_.update({k_: parity[d][k_] for k_ in ("argmax_mismatch_frac", "max_abs_dlogit", "max_reference_margin_at_mismatch")})
                e_["by_fixture"] = {n_.replace("ref_full_vitl16_480x480_", ""): {k_: v[k_] for k_ in ("argmax_mismatch_frac", "max_abs_dlogit",
                                    "max_reference_margin_at_mismatch", "nonfinite_16bit_activations", "max_abs_16bit_activation")}
                                    for n_, v in parity_all.get(d, {}).items()}
            if e_:
                sel[d] = e_
        line = {
            "metric": "images/sec at 480x480, ViT-L/16 + 150 ADE20K labels",
            "value": round(ips, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": chosen, "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: {args.backbone} LSegNet.forward, {args.size}x{args.size}, "
                                   f"K={K} ADE20K labels, text tower recomputed every call",
                       "per_gpu_batch": B, "global_batch": B * world, "labels": K,
                       "parallelism": f"dp{world} (batch sharded, no collectives)"},
            "per_rank_images_per_sec": [round(v, 1) for v in per_rank],
            "parity": parity.get(chosen),
            "miou_parity": "unavailable: no LSeg checkpoint and no ADE20K data in this environment (no network) -- parity is pinned on "
                           "reference-run fixtures with synthetic weights (tests/golden/ref_full_*.pt), mIoU itself is not measured",
            "parity_by_fixture": parity_all.get(chosen),
            "commit": build_id(),
            "dtype_selection": sel,
            "path_tflops": round(world * gf_step / (ms * 1e-3) / 1e3, 2),
            "path_frac_of_mfma_peak": round(gf_step / (ms * 1e-3) / 1e3 / PEAK_BF16_TFLOPS, 4),
            "path_flops_convention": f"executed FLOPs: image tower {GF_IMAGE_EXEC(K):.1f} GF/image (head 1x1 convs and the correlation commuted below the "
                                     f"x2 upsample: -18.75 GF - {(0.05898 - 0.01524) * K:.2f} GF) + text tower at {L_exec} of 77 positions ({gf_text_executed(K, L_exec):.1f} GF/call); "
                                     f"the reference's schedule on the same inputs = {GF_IMAGE(K):.1f} + {GF_TEXT(K):.1f}",
            "path_tflops_reference_algorithm": round(world * gf_step_ref / (ms * 1e-3) / 1e3, 2),
            "engine_forward_ms_hip_events": round(prof["forward"]["total_ms"] / max(1, prof["forward"]["launches"]), 4),
            "roofline": roof,
            "roofline_kernels": kern,
            "selfcheck_batch_vs_single_max_abs": round(selfcheck, 6),
        }
    else:
        line = None

    # ---- extra legs, outside the headline's timed region: per-GPU batch sweep (config 3 runs 4 images per GPU), K = 1000 (config 5),
    # the training step (config 4) and the CPU baseline.  A watchdog guarantees the contract line: if a leg does not come back (a
    # collective that never completes on some rank), rank 0 prints the line without it and every rank leaves.
    import threading

    # the CPU baseline first (bounded on its own: fixed sample counts, the all-cores leg in a child process under a time limit), so that
    # the contract line carries it even if a GPU leg below had to be abandoned
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_baseline(cfg, sd, tok, args.size, args.cpu_threads, args.cpu_samples, args)
        except Exception as e:                           # noqa: BLE001
            line["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}

    def give_up():
        if rank == 0:
            line["extra_legs"] = "abandoned after 420 s"
            print(json.dumps(line), flush=True)
        os._exit(0)

    dog = threading.Timer(420.0, give_up)
    dog.daemon = True
    dog.start()
    sweep, train, k1000, boundary, evalms = None, None, None, None, None
    if not args.no_sweep:
        sweep = {}
        for b in (1, 4, 8, 16):
            if b >= B:
                continue
            tb = D.max_over_ranks(time_forward(eng, x[:b], 10 if b > 1 else 30, 3, sync), device="cuda")
            sweep[str(b)] = round(world * b / tb, 1)
        sweep[str(B)] = round(world * B * args.steps / dt, 1)
        for e in engines.values():
            e.close()
        if headline and B == 36:        # one point ABOVE the headline batch (its own engine: the plan is sized by max_batch), for information
            try:
                e2 = HipEngine(cfg, args.size, args.size, max_batch=2 * B, max_labels=K, image_dtype=chosen)
                e2.load_state_dict(sd)
                e2.set_tokens(tok)
                x2 = torch.cat([x, synthetic_images(B, args.size, args.size, seed=1000 + rank).cuda()], dim=0)
                t2 = D.max_over_ranks(time_forward(e2, x2, 5, 2, sync), device="cuda")
                sweep[str(2 * B)] = round(world * 2 * B / t2, 1)
                e2.close()
                del x2
            except Exception as e:                       # noqa: BLE001
                sweep[str(2 * B)] = f"{type(e).__name__}: {e}"
        if headline:
            try:
                k1000 = k1000_leg(cfg, sd, args.size, 4, sync, D, chosen)
            except Exception as e:                       # noqa: BLE001  (the headline must survive a failing extra leg)
                k1000 = {"error": f"{type(e).__name__}: {e}"}
            try:
                train = train_leg(cfg, sd, tok, args.size, args.train_batch, rank, sync, D, sync_bn=args.train_sync_bn or world == 1)
            except Exception as e:                       # noqa: BLE001
                train = {"error": f"{type(e).__name__}: {e}"}
            if world == 1:
                try:
                    boundary = boundary_leg(args, cfg, sd, labels, chosen, x, sync)
                except Exception as e:                   # noqa: BLE001
                    boundary = {"error": f"{type(e).__name__}: {e}"}
                try:
                    evalms = eval_leg(args, cfg, sd, labels, chosen, sync)
                except Exception as e:                   # noqa: BLE001
                    evalms = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        line["batch_sweep_images_per_sec"] = sweep
        line["config3_per_gpu_batch4_images_per_sec"] = sweep.get("4") if sweep else None
        line["config5_k1000"] = k1000
        line["train_step"] = train
        line["boundary_forward"] = boundary
        line["eval_multiscale"] = evalms
        dog.cancel()
        print(json.dumps(line), flush=True)
    dog.cancel()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
